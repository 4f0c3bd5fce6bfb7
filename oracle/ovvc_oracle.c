/* ovvc_oracle.c -- TEST INFRASTRUCTURE: CPU restatement of OpenVVC's rcn hot path.
 *
 * NOT part of the product.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load liboracle.so, and only as the checker / reported CPU baseline.
 * The product path (libovvc_hip.so) never links, loads or calls anything here.
 *
 * Plain scalar C that consumes the SAME command buffers as the device engine
 * (include/ovvc_hip.h) and host-memory planes, restating the reference algorithm block by
 * block.  Each function cites the reference lines it follows.  Pinned against the compiled
 * reference itself: tests/golden/ fixtures are produced by oracle/ref_harness/gen_golden.c
 * driving the reference's own orchestrators (rcn_tu_st, rcn_mcp_b, ...) from
 * oracle/_ref/libovvcref.so, and tests/test_oracle_golden.py checks this file against them.
 *
 * ONE block shape is PARITY UNPINNED: the 64x2 transform blocks of a 64x8 coding unit split into intra sub-partitions.  The
 * reference's own result for them is undefined (rcn_Xx2_tb, rcn_transform_tree.c:985-1009, reads memory nothing wrote); itx_one
 * below takes the 32 coded columns and nothing else, as H.266 8.7.4 defines it, and tests/spec_isp64x2.py -- a restatement of
 * 8.7.3 / 8.7.4 for that shape -- is what checks it (tests/test_edge_cases_cpu.py).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "ovvc_hip.h"
#include "vvc_tables.h"
#include "vvc_mc_taps.h"

#define BD 10
#define PIX_MAX ((1 << BD) - 1)

typedef struct oracle_pic {      /* host-memory twin of ovhip_pic */
    uint16_t *y, *cb, *cr;
    int32_t w, h, stride_y, stride_c;
} oracle_pic;

static inline int clip3i(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
static inline int clip16(int v) { return clip3i(v, -32768, 32767); }
static inline int clip_bd(int v) { return clip3i(v, 0, PIX_MAX); }
static inline int32_t wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }

static uint16_t *plane_ptr(const oracle_pic *p, int plane, int *stride)
{
    *stride = plane ? p->stride_c : p->stride_y;
    return plane == 0 ? p->y : plane == 1 ? p->cb : p->cr;
}

/* ====================================================================================
 * K1..K4: inverse quantisation, LFNST, inverse transforms, residual add
 * ================================================================================== */

/* transform core lookup: row-major M[k*N + j] (rcn_transform.c:60-61; data_rcn_transform.c) */
static const int8_t *tr_matrix(int type, int log2n)
{
    static const int8_t *const t[3][7] = {
        { 0, 0, ovt_dst7_4, ovt_dst7_8, ovt_dst7_16, ovt_dst7_32, 0 },
        { 0, 0, ovt_dct8_4, ovt_dct8_8, ovt_dct8_16, ovt_dct8_32, 0 },
        { 0, ovt_dct2_2, ovt_dct2_4, ovt_dct2_8, ovt_dct2_16, ovt_dct2_32, ovt_dct2_64 },
    };
    return t[type][log2n];
}

/* derive_nb_rows / derive_nb_cols, rcn_transform_tree.c:78-101 (names as in the reference:
 * "rows" counts coefficient COLUMNS, i.e. the lines of pass 1) */
static int nb_rows_of(uint64_t map)
{
    uint8_t col = (uint8_t)(map | 1);
    for (int s = 8; s < 64; s += 8) col |= (uint8_t)(map >> s);
    return (32 - __builtin_clz((unsigned)col)) << 2;
}
static int nb_cols_of(uint64_t map)
{
    int nz = __builtin_clzll(map | 1);
    return (8 - (nz >> 3)) << 2;
}

/* dequant_tb_4x4(_neg) (rcn_dequant.c:160-312) / dequant_tb(_neg) (rcn_transform_tree.c:104-132) */
/* ov_clip_intp2(v, 16) is SYMMETRIC: [-32767, 32767] (ovutils.h:78-92) */
static inline int clip_intp2_16(int v) { return clip3i(v, -32767, 32767); }
static inline int16_t dequant1(int c, int scale, int shift, int neg)
{
    if (neg) return (int16_t)clip_intp2_16(wmul(c, scale << shift));
    return (int16_t)clip_intp2_16((int32_t)(wmul(c, scale) + ((1 << shift) >> 1)) >> shift);
}

/* one 1-D inverse transform pass as the reference's tr.func[type][log2n] computes it:
 * dst[i*n + j] = clip16((sum_k src[k*stride + i] * M[k][j] + rnd) >> shift), i < lines.
 * All butterflies / fast paths of rcn_transform.c:71-560 are exact integer refactorings of
 * this product (64-point: only input rows 0..31 exist). */
static void tr_pass(const int16_t *src, int16_t *dst, int stride, int type, int log2n, int lines, int shift)
{
    const int n = 1 << log2n;
    const int kmax = n > 32 ? 32 : n;
    const int8_t *m = tr_matrix(type, log2n);
    const int rnd = 1 << (shift - 1);
    for (int i = 0; i < lines; ++i) {
        for (int j = 0; j < n; ++j) {
            int32_t s = 0;
            for (int k = 0; k < kmax; ++k) s += wmul(src[k * stride + i], m[k * n + j]);
            dst[i * n + j] = (int16_t)clip16((int32_t)(s + rnd) >> shift);
        }
    }
}

/* compute_lfnst_{4x4,8x8}{,_tr}, rcn_lfnst.c:41-162 */
static void lfnst_apply(int16_t *coef, int stride, int log2_w, int log2_h, int field)
{
    static const uint8_t scan[16] = { 0, 4, 1, 8, 5, 2, 12, 9, 6, 3, 13, 10, 7, 14, 11, 15 };
    const int is8 = log2_w >= 3 && log2_h >= 3;
    const int set = (field >> 1) & 3, idx = (field >> 3) & 1, tr = (field >> 4) & 1;
    const int8_t *m = is8 ? ovt_lfnst_8x8[set][idx] : ovt_lfnst_4x4[set][idx];
    const int nout = is8 ? 48 : 16;
    const int nin = is8 ? 16 : (log2_w == log2_h ? 8 : 16);
    int16_t sb[16], in[16];
    int out[48];
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) sb[r * 4 + c] = coef[r * stride + c];
    for (int i = 0; i < 16; ++i) in[i] = sb[scan[i]];
    for (int i = 0; i < nout; ++i) {
        int32_t s = 0;
        for (int j = 0; j < nin; ++j) s += in[j] * m[i + j * nout];
        out[i] = clip3i((s + 64) >> 7, -(1 << 15), 1 << 15);
    }
    for (int i = 0; i < nout; ++i) {
        int r, c;
        if (!is8)        { r = i >> 2; c = i & 3; }
        else if (i < 32) { r = i >> 3; c = i & 7; }
        else             { r = 4 + ((i - 32) >> 2); c = i & 3; }
        if (tr) { int t = r; r = c; c = t; }
        coef[r * stride + c] = (int16_t)out[i];
    }
}

/* the eight residual-add variants of rcn_residuals.c:46-222 */
static void residual_apply(uint16_t *dst, int dstride, const int16_t *res, int w, int h, int mode, int scale)
{
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w; ++x) {
            int32_t v = res[y * w + x];
            switch (mode & 3) {
            case OVHIP_RES_SUB:      v = -v; break;
            case OVHIP_RES_ADD_HALF: v = v >> 1; break;
            case OVHIP_RES_SUB_HALF: v = (-v) >> 1; break;
            default: break;
            }
            if (mode & OVHIP_RES_SCALE) {
                int sign = v & (1 << 15);
                v = (clip_bd(abs(v)) * scale + (1 << 10)) >> 11;
                v = clip3i(sign ? -v : v, -(1 << 15), 1 << 15);
            }
            dst[y * dstride + x] = (uint16_t)clip_bd((int32_t)dst[y * dstride + x] + v);
        }
    }
}

/* One transform block.  rcn_residual (rcn_transform_tree.c:415-506), rcn_residual_c (:553-628),
 * transform-skip (:672-716, :1208-1225), ict.add / ict.ict (:1262, :750-756, :846-866). */
static void itx_one(const oracle_pic *pic, const ovhip_tb_cmd *c, const int16_t *arena, const int16_t *lmcs_scales, const oracle_pic *respic)
{
    const int log2_w = c->log2_w, log2_h = c->log2_h;
    const int tb_w = 1 << log2_w, tb_h = 1 << log2_h;
    const int kind = c->kind & 0x3f, raster = !!(c->kind & OVHIP_TB_FLAG_RASTER), bdpcm = !!(c->kind & OVHIP_TB_FLAG_BDPCM);
    const int cw = tb_w > 32 ? 32 : tb_w, ch = tb_h > 32 ? 32 : tb_h;   /* stored coefficient extent */
    const int16_t *src = arena + c->coef_off;
    static _Thread_local int16_t coef[32 * 32], tmp[64 * 64], res[64 * 64];

    memset(coef, 0, sizeof(int16_t) * cw * ch);
    if (raster) {
        /* raster rows of tb_w; what lies outside the coded extent (rows 32.. of a 2x64 block, columns 32.. of a 64x2 one: zero by
         * H.266 8.7.4) is not taken */
        for (int y = 0; y < ch; ++y)
            for (int x = 0; x < cw; ++x) {
                const int16_t v = src[y * tb_w + x];
                coef[y * cw + x] = (kind == OVHIP_TB_TS_RAW || bdpcm) ? v : dequant1(v, c->dq_scale, c->dq_shift, c->dq_neg);
            }
    } else {
        uint64_t map = c->sig_sb_map;
        int n = 0;
        while (map) {
            int b = __builtin_ctzll(map);
            map &= map - 1;
            int16_t *d = coef + (b >> 3) * 4 * cw + (b & 7) * 4;
            for (int r = 0; r < 4; ++r)
                for (int q = 0; q < 4; ++q)
                    d[r * cw + q] = bdpcm ? src[n * 16 + r * 4 + q] : dequant1(src[n * 16 + r * 4 + q], c->dq_scale, c->dq_shift, c->dq_neg);
            ++n;
        }
    }
    if (bdpcm) {
        /* rcn_bdpcm_tb (rcn_transform_tree.c:631-688): accumulate the LEVELS along a row (apply_bdpcm_1) or a column
         * (apply_bdpcm_2) with int16 saturation, then dequant_sb() per 16 samples (none for blocks < 16 samples) */
        if (c->tr_h == 0) {
            for (int y = 0; y < tb_h; ++y)
                for (int x = 1; x < tb_w; ++x) coef[y * tb_w + x] = (int16_t)clip3i(coef[y * tb_w + x - 1] + coef[y * tb_w + x], -(1 << 15), (1 << 15) - 1);
        } else {
            for (int y = 1; y < tb_h; ++y)
                for (int x = 0; x < tb_w; ++x) coef[y * tb_w + x] = (int16_t)clip3i(coef[(y - 1) * tb_w + x] + coef[y * tb_w + x], -(1 << 15), (1 << 15) - 1);
        }
        if (kind == OVHIP_TB_TS)
            for (int i = 0; i < tb_w * tb_h; ++i) coef[i] = dequant1(coef[i], c->dq_scale, c->dq_shift, c->dq_neg);
    }

    if (kind == OVHIP_TB_TS || kind == OVHIP_TB_TS_RAW) {
        memcpy(res, coef, sizeof(int16_t) * tb_w * tb_h);
    } else if (kind == OVHIP_TB_DC) {
        /* inverse_dct_ii_dc, rcn_transform.c:576-598 */
        int v = clip16((((coef[0] + 1) >> 1) + (1 << (14 - BD - 1))) >> (14 - BD));
        for (int i = 0; i < tb_w * tb_h; ++i) res[i] = (int16_t)v;
    } else {
        int nb_row, nb_col;
        if (raster) {
            int l2sw = 1, l2sh = 1;
            if (log2_h > 2) l2sh = 3;
            if (log2_w > 2) l2sw = 3;
            nb_col = (nb_cols_of(c->sig_sb_map) >> 2) << l2sh;
            nb_row = (nb_rows_of(c->sig_sb_map) >> 2) << l2sw;
        } else {
            nb_row = nb_rows_of(c->sig_sb_map);
            nb_col = nb_cols_of(c->sig_sb_map);
        }
        if (c->lfnst & 1) {
            lfnst_apply(coef, cw, log2_w > 5 ? 5 : log2_w, log2_h > 5 ? 5 : log2_h, c->lfnst);
            nb_row = nb_col = 4 << (log2_w >= 3 && log2_h >= 3);
        }
        (void)nb_col;
        if (nb_row > tb_w) nb_row = tb_w;
        memset(tmp, 0, sizeof(int16_t) * tb_w * tb_h);
        if (log2_w == 0) {
            /* rcn_1xX_tb (rcn_transform_tree.c:947-962): ONE vertical transform with the second pass's shift + 1 */
            tr_pass(coef, res, 1, c->tr_v, log2_h, 1, 20 - BD + 1);
        } else if (log2_h == 0) {
            tr_pass(coef, res, 1, c->tr_h, log2_w, 1, 20 - BD + 1);              /* rcn_Xx1_tb (:1011-1027) */
        } else {
            tr_pass(coef, tmp, cw, c->tr_v, log2_h, nb_row, 7);          /* TR_SHIFT_V */
            tr_pass(tmp, res, tb_h, c->tr_h, log2_w, tb_h, 20 - BD);     /* TR_SHIFT_H */
        }
    }

    int stride;
    if (c->res_mode & OVHIP_RES_STORE) {
        /* block of an ordered task: the residual (after the sign / half variant, before chroma scaling) goes to the residual
         * picture, saturated to int16; oracle_intra_tasks adds it to the prediction */
        for (int k = 0; k < 1 + (c->plane2 != 0xff); ++k) {
            const int plane = k ? c->plane2 : c->plane, mode = k ? c->res_mode2 : c->res_mode;
            int16_t *d = (int16_t *)plane_ptr(respic, plane, &stride) + c->y * stride + c->x;
            for (int y = 0; y < tb_h; ++y)
                for (int x = 0; x < tb_w; ++x) {
                    int32_t v = res[y * tb_w + x];
                    switch (mode & 3) {
                    case OVHIP_RES_SUB:      v = -v; break;
                    case OVHIP_RES_ADD_HALF: v = v >> 1; break;
                    case OVHIP_RES_SUB_HALF: v = (-v) >> 1; break;
                    default: break;
                    }
                    d[y * stride + x] = (int16_t)clip16(v);
                }
        }
        return;
    }
    uint16_t *d = plane_ptr(pic, c->plane, &stride) + c->y * stride + c->x;
    /* OVHIP_RES_SCALE_IDX: the scale was derived from the reconstruction (oracle_lmcs_scale) */
    const int scale = (c->res_mode & OVHIP_RES_SCALE_IDX) ? lmcs_scales[c->c_scale] : c->c_scale;
    residual_apply(d, stride, res, tb_w, tb_h, c->res_mode, scale);
    if (c->plane2 != 0xff) {
        d = plane_ptr(pic, c->plane2, &stride) + c->y * stride + c->x;
        residual_apply(d, stride, res, tb_w, tb_h, c->res_mode2, scale);
    }
}

/* respic: int16 planes of the picture's geometry receiving the OVHIP_RES_STORE blocks (NULL: none expected) */
void oracle_itx_res(const oracle_pic *pic, const ovhip_tb_cmd *cmds, uint32_t n, const int16_t *arena, const int16_t *lmcs_scales,
                    const oracle_pic *respic)
{
    for (uint32_t i = 0; i < n; ++i) itx_one(pic, &cmds[i], arena, lmcs_scales, respic);
}

void oracle_itx_ex(const oracle_pic *pic, const ovhip_tb_cmd *cmds, uint32_t n, const int16_t *arena, const int16_t *lmcs_scales)
{
    oracle_itx_res(pic, cmds, n, arena, lmcs_scales, NULL);
}

void oracle_itx(const oracle_pic *pic, const ovhip_tb_cmd *cmds, uint32_t n, const int16_t *arena)
{
    oracle_itx_ex(pic, cmds, n, arena, NULL);
}

/* ====================================================================================
 * K11: LMCS chroma-scale derivation and inverse luma mapping
 * ================================================================================== */

/* rcn_lmcs_compute_chroma_scale + lmcs_compute_luma_average + get_bwd_idx (rcn_lmcs.c:83-93, :204-350) */
static void lmcs_scale_regions(const oracle_pic *pic, const ovhip_lmcs_region *regs, uint32_t n, const ovhip_lmcs_luts *luts,
                               int16_t *scales, int ordered_too);

/* the regions marked `ordered` are left to the ordered pass (oracle_intra_tasks) */
void oracle_lmcs_scale(const oracle_pic *pic, const ovhip_lmcs_region *regs, uint32_t n,
                       const ovhip_lmcs_luts *luts, int16_t *scales)
{
    lmcs_scale_regions(pic, regs, n, luts, scales, 0);
}

static void lmcs_scale_regions(const oracle_pic *pic, const ovhip_lmcs_region *regs, uint32_t n, const ovhip_lmcs_luts *luts,
                               int16_t *scales, int ordered_too)
{
    for (uint32_t r = 0; r < n; ++r) {
        const ovhip_lmcs_region *g = &regs[r];
        if (g->ordered && !ordered_too) continue;
        const uint16_t *src = pic->y + g->y * pic->stride_y + g->x;
        uint32_t s1 = 0, s2 = 0, s3 = 0, s4 = 0;
        int nb_abv = 0, nb_lft = 0, nb_units, log2_nb = 0;
        const uint16_t *p = src - pic->stride_y;
        for (int u = 0; u < g->n_abv; ++u, p += 4, ++nb_abv) { s1 += p[0]; s2 += p[1]; s3 += p[2]; s4 += p[3]; }
        if (nb_abv) { uint32_t pad = p[-1] * (16 - nb_abv); s1 += pad; s2 += pad; s3 += pad; s4 += pad; nb_abv = 16; }
        p = src - 1;
        for (int u = 0; u < g->n_lft; ++u, p += 4 * pic->stride_y, ++nb_lft) {
            s1 += p[0]; s2 += p[pic->stride_y]; s3 += p[2 * pic->stride_y]; s4 += p[3 * pic->stride_y];
        }
        if (nb_lft) { uint32_t pad = p[-pic->stride_y] * (16 - nb_lft); s1 += pad; s2 += pad; s3 += pad; s4 += pad; nb_lft = 16; }
        nb_units = nb_abv + nb_lft;
        while (nb_units) { ++log2_nb; nb_units >>= 1; }
        uint32_t avg = log2_nb ? ((s1 + s2) + (s3 + s4) + (1u << log2_nb)) >> (log2_nb + 1) : 1u << (BD - 1);
        int idx = luts->min_idx;
        for (; idx < luts->max_idx; ++idx)
            if (avg < luts->wnd_bnd[idx + 1]) break;
        if (idx > 15) idx = 15;
        int32_t wnd_sz = (int32_t)luts->wnd_bnd[idx + 1] - (int32_t)luts->wnd_bnd[idx];
        scales[r] = (int16_t)(wnd_sz == 0 ? 1 << 11 : (1 << (BD - 4 + 11)) / (wnd_sz + luts->crs_offset));
    }
}

/* lmcs_reshape_backward over the picture (rcn_lmcs.c:273-301; slicedec.c:746-750) */
void oracle_lmcs_inverse(const oracle_pic *pic, const uint16_t *bwd_lut)
{
    for (int y = 0; y < pic->h; ++y)
        for (int x = 0; x < pic->w; ++x) pic->y[y * pic->stride_y + x] = bwd_lut[pic->y[y * pic->stride_y + x] & PIX_MAX];
}

/* ====================================================================================
 * K5/K6/K11: motion compensation, uni / bi / BCW, luma + chroma, LMCS forward reshape
 * ================================================================================== */

/* Reference sample fetch.  Plain MC: coordinates clamped to the picture = emulate_block_border()
 * (rcn_inter.c:148-225).  DMVR: the (w+7)x(h+7) [chroma (w+3)x(h+3)] window fetched at the INITIAL
 * integer position is first replicated 2 samples outwards (padd_dmvr / padd_dmvr_c,
 * rcn_inter.c:326-378) and the refined block reads THAT, so offsets are clamped to the window
 * (win = 1) before the picture clamp. */
typedef struct sampler {
    const uint16_t *ref; int rstride, rw, rh;
    int px, py;          /* window anchor: block position + initial integer MV             */
    int dx, dy;          /* integer displacement of the block actually predicted (DMVR)    */
    int win, lo, hi_x, hi_y;
} sampler;

static inline int smp(const sampler *s, int i, int j)
{
    i += s->dx; j += s->dy;
    if (s->win) { i = clip3i(i, s->lo, s->hi_x); j = clip3i(j, s->lo, s->hi_y); }
    return s->ref[clip3i(s->py + j, 0, s->rh - 1) * s->rstride + clip3i(s->px + i, 0, s->rw - 1)];
}

/* 14-bit intermediate prediction of one block, exactly as put_vvc_{pel,qpel,epel}*_{h,v,hv}
 * compute it (rcn_mc.c:402-420, :903-985, :1188-1270).  The four reference variants are one
 * separable filter whose integer-position row is the identity tap (see vvc_mc_taps.h):
 *   t = F_h(src) >> (BD-8);  P = F_v(t) >> 6. */
static void predict14(int16_t *out, int ow, const sampler *s, int w, int h, const int8_t *fh, const int8_t *fv, int ntaps)
{
    const int before = ntaps == 8 ? 3 : 1;
    int32_t tmp[(16 + 7) * 16];
    for (int y = 0; y < h + ntaps - 1; ++y) {
        for (int x = 0; x < w; ++x) {
            int32_t a = 0;
            for (int t = 0; t < ntaps; ++t) a += fh[t] * (int32_t)smp(s, x + t - before, y - before);
            tmp[y * 16 + x] = (int16_t)(a >> (BD - 8));
        }
    }
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w; ++x) {
            int32_t a = 0;
            for (int t = 0; t < ntaps; ++t) a += fv[t] * tmp[(y + t) * 16 + x];
            out[y * ow + x] = (int16_t)(a >> 6);
        }
    }
}

static void luma_filters(const ovhip_mc_unit *u, int mvx, int mvy, const int8_t **fh, const int8_t **fv, int *ext_x, int *ext_y)
{
    int fx = mvx & 15, fy = mvy & 15;
    if (u->flags & OVHIP_MC_FILT_4x4) { *fh = ovt_mc_luma4[fx]; *fv = ovt_mc_luma4[fy]; }
    else {
        if (u->flags & OVHIP_MC_HPEL_FILT) { if (fx == 8) fx = 16; if (fy == 8) fy = 16; }   /* rcn_inter.c:572-577 */
        *fh = ovt_mc_luma[fx]; *fv = ovt_mc_luma[fy];
    }
    if (ext_x) { *ext_x = (fx >> 3) != 0; *ext_y = (fy >> 3) != 0; }
}

static sampler plane_sampler(const oracle_pic *rp, int plane, int px, int py)
{
    sampler s;
    memset(&s, 0, sizeof(s));
    s.ref = plane_ptr(rp, plane, &s.rstride);
    s.rw = rp->w >> (plane != 0); s.rh = rp->h >> (plane != 0);
    s.px = px; s.py = py;
    return s;
}

static int gpm_weight(uint32_t aux, int x, int y)
{
    const int k = (int16_t)(aux & 0xffff), a = (int8_t)((aux >> 16) & 0xff), b = (int8_t)(aux >> 24);
    return clip3i((k + a * x + b * y) >> 3, 0, 8);
}

static int bi_combine(const ovhip_mc_unit *u, int p0, int p1)
{
    if (u->dir != 3) return clip_bd(((u->dir == 1 ? p0 : p1) + 8) >> 4);          /* uni: rcn_mc.c:448-533 */
    if (u->w0 == 4 && u->w1 == 4) return clip_bd((p0 + p1 + 16) >> 5);            /* bi: rcn_mc.c:422-444, :987-1098 */
    return clip_bd((p1 * u->w1 + p0 * u->w0 + 64) >> 7);                          /* BCW: rcn_mc.c:1480-1610 */
}

static void mc_plane(const oracle_pic *dst, const oracle_pic *refs, const ovhip_mc_unit *u, int plane,
                     const uint16_t *lmcs_fwd, const oracle_pic *intra)
{
    const int c = plane != 0;
    const int w = u->w >> c, h = u->h >> c;
    const int x = u->x >> c, y = u->y >> c;
    int16_t p[2][16 * 16];
    int dstride;
    uint16_t *d = plane_ptr(dst, plane, &dstride) + y * dstride + x;

    for (int l = 0; l < 2; ++l) {
        if (!(u->dir & (1 << l))) continue;
        const oracle_pic *rp = &refs[l ? u->ref1 : u->ref0];
        int mvx = l ? u->mv1x : u->mv0x, mvy = l ? u->mv1y : u->mv0y;
        const int8_t *fh, *fv;
        if (!c) {
            sampler s = plane_sampler(rp, 0, x + (mvx >> 4), y + (mvy >> 4));
            luma_filters(u, mvx, mvy, &fh, &fv, NULL, NULL);
            predict14(p[l], 16, &s, w, h, fh, fv, 8);
        } else {
            sampler s = plane_sampler(rp, plane, x + (mvx >> 5), y + (mvy >> 5));
            predict14(p[l], 16, &s, w, h, ovt_mc_chroma[mvx & 31], ovt_mc_chroma[mvy & 31], 4);
        }
    }
    for (int j = 0; j < h; ++j) {
        for (int i = 0; i < w; ++i) {
            int v = bi_combine(u, p[0][j * 16 + i], p[1][j * 16 + i]);
            if (u->flags & OVHIP_MC_GPM) {                                        /* put_weighted_gpm_bi_pixels, rcn_mc.c:1630-1655 */
                const int wgt = gpm_weight(u->aux, i << c, j << c);
                v = clip_bd((p[1][j * 16 + i] * (8 - wgt) + p[0][j * 16 + i] * wgt + 64) >> 7);
            }
            if (!c && (u->flags & OVHIP_MC_LMCS) && lmcs_fwd) v = lmcs_fwd[v & PIX_MAX]; /* rcn_lmcs.c:275-295 */
            if (!(u->flags & OVHIP_MC_GPM) && u->aux && intra && !(c && (u->aux & 0x100))) {
                /* fused CIIP blend: rcn_ciip_weighted_sum + put_weighted_ciip_pixels (rcn_inter.c:2968-3009, rcn_mc.c:1611-1628) */
                int is;
                const uint16_t *ip = plane_ptr(intra, plane, &is) + y * is + x;
                const int wt = u->aux & 7;
                v = clip_bd((ip[j * is + i] * wt + v * (4 - wt) + 2) >> 2);
            }
            d[j * dstride + i] = (uint16_t)v;
        }
    }
}

/* ---- K8: BDOF on one <=16x16 luma block (rcn_bdof_mcp_l rcn_inter.c:1136-1250; rcn_prof_bdof.c:303-490).
 * p0/p1: 14-bit predictions (stride 16); s0/s1: samplers positioned on the predicted blocks. */
#define BDOF_RS 18
static int floor_log2u(uint32_t v) { int r = -1; while (v) { v >>= 1; ++r; } return r; }

static void bdof_block(uint16_t *d, int dstride, const int16_t *p0, const int16_t *p1, const sampler *s0, const sampler *s1,
                       const int ext[2][2], int w, int h)
{
    int16_t R[2][BDOF_RS * BDOF_RS], GX[2][BDOF_RS * BDOF_RS], GY[2][BDOF_RS * BDOF_RS];
    const int16_t *p[2] = { p0, p1 };
    const sampler *s[2] = { s0, s1 };
    for (int l = 0; l < 2; ++l) {
        int16_t *r = R[l];
        /* interior = prediction; ring = integer reference samples << 4 (extend_bdof_buff, rcn_prof_bdof.c:303-352) */
        for (int j = 0; j < h + 2; ++j)
            for (int i = 0; i < w + 2; ++i) {
                int in = i >= 1 && i <= w && j >= 1 && j <= h;
                r[j * BDOF_RS + i] = in ? p[l][(j - 1) * 16 + i - 1]
                                        : (int16_t)(smp(s[l], i - 1 + ext[l][0], j - 1 + ext[l][1]) << (14 - BD));
            }
        /* gradients of the interior (compute_prof_grad, rcn_prof_bdof.c:152-172) */
        for (int j = 1; j <= h; ++j)
            for (int i = 1; i <= w; ++i) {
                int16_t gy = (int16_t)((r[(j + 1) * BDOF_RS + i] - (1 << 13)) >> 6);
                gy -= (int16_t)((r[(j - 1) * BDOF_RS + i] - (1 << 13)) >> 6);
                int16_t gx = (int16_t)((r[j * BDOF_RS + i + 1] - (1 << 13)) >> 6);
                gx -= (int16_t)((r[j * BDOF_RS + i - 1] - (1 << 13)) >> 6);
                GX[l][j * BDOF_RS + i] = gx; GY[l][j * BDOF_RS + i] = gy;
            }
        /* replicate gradients AND predictions one sample outwards (extend_bdof_grad, rcn_inter.c:840-869) */
        int16_t *planes[3] = { GX[l], GY[l], r };
        for (int k = 0; k < 3; ++k) {
            int16_t *a = planes[k];
            for (int j = 1; j <= h; ++j) { a[j * BDOF_RS] = a[j * BDOF_RS + 1]; a[j * BDOF_RS + w + 1] = a[j * BDOF_RS + w]; }
            for (int i = 0; i < w + 2; ++i) { a[i] = a[BDOF_RS + i]; a[(h + 1) * BDOF_RS + i] = a[h * BDOF_RS + i]; }
        }
    }
    for (int sy = 0; sy < h; sy += 4) {
        for (int sx = 0; sx < w; sx += 4) {
            /* derive_bdof_weights (rcn_prof_bdof.c:355-428): 6x6 window whose top-left is the padded (sx, sy) */
            int sum_ax = 0, sum_ay = 0, sum_dx = 0, sum_dy = 0, sum_xy = 0, wx = 0, wy = 0;
            for (int j = 0; j < 6; ++j)
                for (int i = 0; i < 6; ++i) {
                    int o = (sy + j) * BDOF_RS + sx + i;
                    int32_t ax = (GX[0][o] + GX[1][o]) >> 1, ay = (GY[0][o] + GY[1][o]) >> 1;
                    int32_t dr = ((R[1][o] - (1 << 13)) >> 4) - ((R[0][o] - (1 << 13)) >> 4);
                    sum_ax += abs(ax); sum_ay += abs(ay);
                    sum_xy += ay < 0 ? -ax : (ay == 0 ? 0 : ax);
                    sum_dx += ax < 0 ? -dr : (ax == 0 ? 0 : dr);
                    sum_dy += ay < 0 ? -dr : (ay == 0 ? 0 : dr);
                }
            if (sum_ax) wx = clip3i((sum_dx * 4) >> floor_log2u(sum_ax), -15, 15);
            if (sum_ay) {
                int x_off = 0;
                if (wx) {
                    int high = sum_xy >> 12, low = sum_xy & 4095;
                    x_off = (((wx * high) * 4096) + (wx * low)) >> 1;
                }
                wy = clip3i(((sum_dy * 4) - x_off) >> floor_log2u(sum_ay), -15, 15);
            }
            /* rcn_apply_bdof_subblock (rcn_prof_bdof.c:59-103) */
            for (int j = 0; j < 4; ++j)
                for (int i = 0; i < 4; ++i) {
                    int o = (sy + j + 1) * BDOF_RS + sx + i + 1;
                    int32_t b = wx * (GX[0][o] - GX[1][o]) + wy * (GY[0][o] - GY[1][o]);
                    int16_t v = (int16_t)((R[0][o] + R[1][o] + b + 16) >> 5);
                    d[(sy + j) * dstride + sx + i] = (uint16_t)clip_bd(v);
                }
        }
    }
}

/* ---- K7: DMVR (rcn_dmvr_mv_refine rcn_inter.c:872-1126) ---- */
static void clip_mv(int pos_x, int pos_y, int pic_w, int pic_h, int pb_w, int pb_h, int *mvx, int *mvy)
{   /* rcn_inter.c:96-109 */
    *mvx = clip3i(*mvx, -((pb_w + 3 + pos_x) << 4), (pic_w + 2 - pos_x) << 4);
    *mvy = clip3i(*mvy, -((pb_h + 3 + pos_y) << 4), (pic_h + 2 - pos_y) << 4);
}

/* put_vvc_{pel_bilinear_pixels,qpel_bilinear_h,_v,_hv} (rcn_mc.c:788-899) on the (w+4)x(h+4) block that
 * starts 2 samples up-left of the prediction block; out stride 20 */
static void dmvr_bilinear(int16_t *out, const sampler *s, int w, int h, int fx, int fy)
{
    int16_t t[21 * 20];
    for (int j = 0; j < h + 5; ++j)
        for (int i = 0; i < w + 4; ++i)
            t[j * 20 + i] = fx ? (int16_t)(((16 - fx) * smp(s, i - 2, j - 2) + fx * smp(s, i - 1, j - 2) + 8) >> 4)
                               : (int16_t)smp(s, i - 2, j - 2);
    for (int j = 0; j < h + 4; ++j)
        for (int i = 0; i < w + 4; ++i)
            out[j * 20 + i] = fy ? (int16_t)(((16 - fy) * t[j * 20 + i] + fy * t[(j + 1) * 20 + i] + 8) >> 4) : t[j * 20 + i];
}

/* rcn_dmvr_sad_8/_16 (rcn_inter.c:614-664): every second row */
static uint64_t dmvr_sad(const int16_t *b0, const int16_t *b1, int w, int h)
{
    uint64_t sum = 0;
    for (int j = 0; j < h; j += 2)
        for (int i = 0; i < w; ++i) sum += (uint64_t)abs(b0[j * 20 + i] - b1[j * 20 + i]);
    return sum;
}

static int32_t div_for_maxq7(int64_t num, int64_t den)
{   /* rcn_inter.c:758-797: 3-bit restoring division, result in 1/16 pel, |q| <= 7 */
    int32_t sign = 0, q = 0;
    if (num < 0) { sign = 1; num = -num; }
    den <<= 3;
    if (num >= den) { num -= den; q++; }
    q <<= 1; den >>= 1;
    if (num >= den) { num -= den; q++; }
    q <<= 1;
    if (num >= (den >> 1)) q++;
    return sign ? -q : q;
}

/* One BDOF and/or DMVR unit (luma <= 16x16 + chroma): rcn_bdof_mcp_l (rcn_inter.c:1136-1250) followed
 * by the chroma of rcn_mcp_b_c, or rcn_dmvr_mv_refine (rcn_inter.c:872-1126). */
static void mc_refined_unit(const oracle_pic *dst, const oracle_pic *refs, const ovhip_mc_unit *u,
                            const uint16_t *lmcs_fwd, int32_t *mv_out)
{
    const int w = u->w, h = u->h;
    const int dmvr = (u->flags & OVHIP_MC_DMVR) != 0;
    int use_bdof = (u->flags & OVHIP_MC_BDOF) != 0;
    int mv[2][2] = { { u->mv0x, u->mv0y }, { u->mv1x, u->mv1y } };
    int ini[2][2];
    sampler sl[2];
    memcpy(ini, mv, sizeof(ini));

    for (int l = 0; l < 2; ++l) {
        const oracle_pic *rp = &refs[l ? u->ref1 : u->ref0];
        int cx = mv[l][0], cy = mv[l][1];
        if (dmvr) clip_mv(u->x, u->y, rp->w, rp->h, w, h, &cx, &cy);   /* derive_dmvr_ref_buf_y, rcn_inter.c:431-471 */
        sl[l] = plane_sampler(rp, 0, u->x + (cx >> 4), u->y + (cy >> 4));
        if (dmvr) { sl[l].win = 1; sl[l].lo = -3; sl[l].hi_x = w + 3; sl[l].hi_y = h + 3; }
    }

    if (dmvr) {
        int16_t B[2][24 * 20];
        for (int l = 0; l < 2; ++l) dmvr_bilinear(B[l], &sl[l], w, h, ini[l][0] & 15, ini[l][1] & 15);
        uint64_t sad_c = dmvr_sad(B[0] + 2 * 20 + 2, B[1] + 2 * 20 + 2, w, h);
        uint64_t min_cost = sad_c - (sad_c >> 2);
        if (min_cost >= (uint64_t)(w * h)) {
            uint64_t sad[25], best = (uint64_t)-1;
            int idx = 12;
            sad[12] = min_cost;
            for (int k = 0; k < 25; ++k) {
                if (k == 12) continue;
                int dx = k % 5 - 2, dy = k / 5 - 2;                         /* dmvr_mv_x/_y, rcn_inter.c:63-87 */
                sad[k] = dmvr_sad(B[0] + (2 + dy) * 20 + 2 + dx, B[1] + (2 - dy) * 20 + 2 - dx, w, h);
            }
            for (int k = 0; k < 25; ++k)                                    /* dmvr_compute_sads_*, rcn_inter.c:666-754 */
                if (sad[k] < best || (k == 12 && sad[k] <= best)) { best = sad[k]; idx = k; }
            int dh = (idx % 5 - 2) << 4, dv = (idx / 5 - 2) << 4;
            min_cost = sad[idx];
            if (abs(dh) != 32 && abs(dv) != 32) {                           /* refine_mv, rcn_inter.c:799-833 */
                uint64_t s0 = sad[idx], s1 = sad[idx - 1], s3 = sad[idx + 1], s2 = sad[idx - 5], s4 = sad[idx + 5];
                int64_t den_h = (int64_t)(s1 + s3 - (s0 << 1)), den_v = (int64_t)(s2 + s4 - (s0 << 1));
                if (den_h) dh += (s1 != s0 && s3 != s0) ? div_for_maxq7((int64_t)(s1 << 4) - (int64_t)(s3 << 4), den_h) : (s1 == s0 ? -8 : 8);
                if (den_v) dv += (s2 != s0 && s4 != s0) ? div_for_maxq7((int64_t)((s2 - s4) << 4), den_v) : (s2 == s0 ? -8 : 8);
            }
            mv[0][0] = clip3i(mv[0][0] + dh, -(1 << 17), (1 << 17) - 1); mv[0][1] = clip3i(mv[0][1] + dv, -(1 << 17), (1 << 17) - 1);
            mv[1][0] = clip3i(mv[1][0] - dh, -(1 << 17), (1 << 17) - 1); mv[1][1] = clip3i(mv[1][1] - dv, -(1 << 17), (1 << 17) - 1);
        }
        if (use_bdof && min_cost < (uint64_t)(2 * w * h)) use_bdof = 0;
    }
    if (mv_out) { mv_out[0] = mv[0][0]; mv_out[1] = mv[0][1]; mv_out[2] = mv[1][0]; mv_out[3] = mv[1][1]; }

    if (!(u->flags & OVHIP_MC_NO_LUMA)) {
        int16_t p[2][16 * 16];
        int ext[2][2];
        for (int l = 0; l < 2; ++l) {
            const int8_t *fh, *fv;
            luma_filters(u, mv[l][0], mv[l][1], &fh, &fv, &ext[l][0], &ext[l][1]);
            sl[l].dx = (mv[l][0] >> 4) - (ini[l][0] >> 4);
            sl[l].dy = (mv[l][1] >> 4) - (ini[l][1] >> 4);
            predict14(p[l], 16, &sl[l], w, h, fh, fv, 8);
        }
        uint16_t *d = dst->y + u->y * dst->stride_y + u->x;
        if (use_bdof) bdof_block(d, dst->stride_y, p[0], p[1], &sl[0], &sl[1], ext, w, h);
        else
            for (int j = 0; j < h; ++j)
                for (int i = 0; i < w; ++i) d[j * dst->stride_y + i] = (uint16_t)clip_bd((p[0][j * 16 + i] + p[1][j * 16 + i] + 16) >> 5);
        if ((u->flags & OVHIP_MC_LMCS) && lmcs_fwd)
            for (int j = 0; j < h; ++j)
                for (int i = 0; i < w; ++i) d[j * dst->stride_y + i] = lmcs_fwd[d[j * dst->stride_y + i] & PIX_MAX];
    }
    if (!(u->flags & OVHIP_MC_NO_CHROMA)) {
        const int wc = w >> 1, hc = h >> 1;
        for (int plane = 1; plane < 3; ++plane) {
            int16_t p[2][16 * 16];
            for (int l = 0; l < 2; ++l) {
                const oracle_pic *rp = &refs[l ? u->ref1 : u->ref0];
                int cx = ini[l][0], cy = ini[l][1];
                if (dmvr) clip_mv(u->x, u->y, rp->w, rp->h, w, h, &cx, &cy);   /* derive_dmvr_ref_buf_c, rcn_inter.c:473-517 */
                sampler s = plane_sampler(rp, plane, (u->x >> 1) + (cx >> 5), (u->y >> 1) + (cy >> 5));
                if (dmvr) { s.win = 1; s.lo = -1; s.hi_x = wc + 1; s.hi_y = hc + 1; }
                s.dx = (mv[l][0] >> 5) - (ini[l][0] >> 5);
                s.dy = (mv[l][1] >> 5) - (ini[l][1] >> 5);
                predict14(p[l], 16, &s, wc, hc, ovt_mc_chroma[mv[l][0] & 31], ovt_mc_chroma[mv[l][1] & 31], 4);
            }
            int dstride;
            uint16_t *d = plane_ptr(dst, plane, &dstride) + (u->y >> 1) * dstride + (u->x >> 1);
            for (int j = 0; j < hc; ++j)
                for (int i = 0; i < wc; ++i) d[j * dstride + i] = (uint16_t)clip_bd((p[0][j * 16 + i] + p[1][j * 16 + i] + 16) >> 5);
        }
    }
}

/* rcn_mcp_l/_c, rcn_motion_compensation_b_l/_c (rcn_inter.c:520-602, :1391-1554, :1822-1904);
 * units flagged OVHIP_MC_BDOF / OVHIP_MC_DMVR go through mc_refined_unit.  mv_out (may be NULL):
 * 4 int32 per unit, the motion vectors finally used (DMVR write-back for TMVP,
 * vcl_coding_unit.c:2621-2645). */
void oracle_mc_full(const oracle_pic *dst, const oracle_pic *refs, uint32_t n_refs,
                    const ovhip_mc_unit *units, uint32_t n, const uint16_t *lmcs_fwd, int32_t *mv_out, const oracle_pic *intra)
{
    (void)n_refs;
    for (uint32_t i = 0; i < n; ++i) {
        const ovhip_mc_unit *u = &units[i];
        if (u->flags & (OVHIP_MC_BDOF | OVHIP_MC_DMVR)) { mc_refined_unit(dst, refs, u, lmcs_fwd, mv_out ? mv_out + 4 * i : NULL); continue; }
        if (mv_out) { mv_out[4 * i] = u->mv0x; mv_out[4 * i + 1] = u->mv0y; mv_out[4 * i + 2] = u->mv1x; mv_out[4 * i + 3] = u->mv1y; }
        if (!(u->flags & OVHIP_MC_NO_LUMA)) mc_plane(dst, refs, u, 0, lmcs_fwd, intra);
        if (!(u->flags & OVHIP_MC_NO_CHROMA)) { mc_plane(dst, refs, u, 1, lmcs_fwd, intra); mc_plane(dst, refs, u, 2, lmcs_fwd, intra); }
    }
}

void oracle_mc_ex(const oracle_pic *dst, const oracle_pic *refs, uint32_t n_refs,
                  const ovhip_mc_unit *units, uint32_t n, const uint16_t *lmcs_fwd, int32_t *mv_out)
{
    oracle_mc_full(dst, refs, n_refs, units, n, lmcs_fwd, mv_out, NULL);
}

void oracle_mc(const oracle_pic *dst, const oracle_pic *refs, uint32_t n_refs,
               const ovhip_mc_unit *units, uint32_t n, const uint16_t *lmcs_fwd)
{
    oracle_mc_ex(dst, refs, n_refs, units, n, lmcs_fwd, NULL);
}

/* ---- K9: affine sub-block prediction + PROF ----
 * One 4x4 luma sub-block: rcn_mcp_b_l(2,2) -> rcn_mcp_l / rcn_motion_compensation_b_l, or
 * rcn_prof_mcp_b_l -> rcn_prof_mcp_l / rcn_prof_motion_compensation_b_l (rcn_inter.c:1252-1386,
 * :1631-1722, :2864-2918). */
static void prof_refine(int16_t *p /* 4x4, stride 16 */, const sampler *s, int ext_x, int ext_y,
                        const int16_t *dmv_h, const int16_t *dmv_v)
{
    int16_t t[6 * 6];
    /* extend_prof_buff (rcn_prof_bdof.c:174-223): ring of integer reference samples << 4 */
    for (int j = 0; j < 6; ++j)
        for (int i = 0; i < 6; ++i) {
            int in = i >= 1 && i <= 4 && j >= 1 && j <= 4;
            t[j * 6 + i] = in ? p[(j - 1) * 16 + i - 1] : (int16_t)(smp(s, i - 1 + ext_x, j - 1 + ext_y) << (14 - BD));
        }
    for (int j = 1; j <= 4; ++j)
        for (int i = 1; i <= 4; ++i) {
            /* compute_prof_grad (rcn_prof_bdof.c:152-172) */
            int16_t gy = (int16_t)((t[(j + 1) * 6 + i] - (1 << 13)) >> 6);
            gy -= (int16_t)((t[(j - 1) * 6 + i] - (1 << 13)) >> 6);
            int16_t gx = (int16_t)((t[j * 6 + i + 1] - (1 << 13)) >> 6);
            gx -= (int16_t)((t[j * 6 + i - 1] - (1 << 13)) >> 6);
            /* rcn_prof (rcn_prof_bdof.c:225-280) */
            int idx = (j - 1) * 4 + i - 1;
            int32_t add = clip3i(dmv_h[idx] * gx + dmv_v[idx] * gy, -(1 << 13), (1 << 13) - 1);
            p[(j - 1) * 16 + i - 1] = (int16_t)(t[j * 6 + i] + add);
        }
}

/* rcn_affine_mcp_b_l / rcn_affine_prof_mcp_b_l / rcn_affine_mcp_b_c (drv_affine_mvp.c:3264-3411) */
void oracle_mca(const oracle_pic *dst, const oracle_pic *refs, uint32_t n_refs,
                const ovhip_aff_unit *units, uint32_t n, const int32_t *side, const uint16_t *lmcs_fwd)
{
    (void)n_refs;
    for (uint32_t ui = 0; ui < n; ++ui) {
        const ovhip_aff_unit *u = &units[ui];
        const int32_t *mvs = side + u->side_off;
        const int16_t *prof = (const int16_t *)(side + u->prof_off);
        const int nsx = u->w >> 2;
        for (int sb = 0; sb < (u->w >> 2) * (u->h >> 2); ++sb, mvs += 4) {
            const int x = u->x + 4 * (sb % nsx), y = u->y + 4 * (sb / nsx);
            int dir = u->dir;
            if ((u->ident_l >> sb) & 1) dir = 2;
            ovhip_mc_unit m;
            memset(&m, 0, sizeof(m));
            m.dir = (uint8_t)dir; m.w0 = u->w0; m.w1 = u->w1;
            int16_t p[2][16 * 16];
            for (int l = 0; l < 2; ++l) {
                if (!(dir & (1 << l))) continue;
                const oracle_pic *rp = &refs[l ? u->ref1 : u->ref0];
                const int mvx = mvs[2 * l], mvy = mvs[2 * l + 1], fx = mvx & 15, fy = mvy & 15;
                sampler s = plane_sampler(rp, 0, x + (mvx >> 4), y + (mvy >> 4));
                predict14(p[l], 16, &s, 4, 4, ovt_mc_luma4[fx], ovt_mc_luma4[fy], 8);
                if ((u->flags & OVHIP_AFF_PROF) && (dir != 3 || ((u->prof_dir >> l) & 1)))
                    prof_refine(p[l], &s, fx >> 3, fy >> 3, prof + 32 * l, prof + 32 * l + 16);
            }
            for (int j = 0; j < 4; ++j)
                for (int i = 0; i < 4; ++i) {
                    int v = bi_combine(&m, p[0][j * 16 + i], p[1][j * 16 + i]);
                    if ((u->flags & OVHIP_AFF_LMCS) && lmcs_fwd) v = lmcs_fwd[v & PIX_MAX];
                    dst->y[(y + j) * dst->stride_y + x + i] = (uint16_t)v;
                }
        }
        if (u->flags & OVHIP_AFF_NO_CHROMA) continue;
        const int ncx = u->w >> 3;
        for (int cbk = 0; cbk < (u->w >> 3) * (u->h >> 3); ++cbk, mvs += 4) {
            ovhip_mc_unit m;
            memset(&m, 0, sizeof(m));
            m.x = (uint16_t)(u->x + 8 * (cbk % ncx)); m.y = (uint16_t)(u->y + 8 * (cbk / ncx));
            m.w = m.h = 8;
            m.dir = ((u->ident_c >> cbk) & 1) ? 2 : u->dir;
            m.flags = OVHIP_MC_NO_LUMA;
            m.ref0 = u->ref0; m.ref1 = u->ref1; m.w0 = u->w0; m.w1 = u->w1;
            m.mv0x = mvs[0]; m.mv0y = mvs[1]; m.mv1x = mvs[2]; m.mv1y = mvs[3];
            mc_plane(dst, refs, &m, 1, NULL, NULL);
            mc_plane(dst, refs, &m, 2, NULL, NULL);
        }
    }
}

/* ---- K10: CIIP blend (rcn_ciip_weighted_sum rcn_inter.c:2968-3009, put_weighted_ciip_pixels rcn_mc.c:1611-1628) ---- */
void oracle_ciip(const oracle_pic *dst, const oracle_pic *intra, const ovhip_ciip_unit *units, uint32_t n)
{
    for (uint32_t k = 0; k < n; ++k) {
        const ovhip_ciip_unit *u = &units[k];
        for (int plane = 0; plane < 3; ++plane) {
            const int c = plane != 0;
            if (c && u->chroma_inter) continue;
            int ds, is;
            uint16_t *d = plane_ptr(dst, plane, &ds) + (u->y >> c) * ds + (u->x >> c);
            const uint16_t *s = plane_ptr(intra, plane, &is) + (u->y >> c) * is + (u->x >> c);
            for (int j = 0; j < (1 << u->log2_h) >> c; ++j)
                for (int i = 0; i < (1 << u->log2_w) >> c; ++i)
                    d[j * ds + i] = (uint16_t)clip_bd((s[j * is + i] * u->wt + d[j * ds + i] * (4 - u->wt) + 2) >> 2);
        }
    }
}

/* ====================================================================================
 * K12: deblocking filter on picture-level edge planes
 * ================================================================================== */
#include "vvc_dbf_tables.h"

typedef struct { int tc, beta; } dbf_lim;

/* compute_dbf_limits, rcn_df.c:171-188 (BITDEPTH 10) */
static dbf_lim dbf_limits(int qp, int bs, int tc_off, int beta_off)
{
    dbf_lim l;
    l.tc = ovt_dbf_tc[clip3i(qp + 2 * (bs - 1) + tc_off, 0, 66)];
    l.beta = ovt_dbf_beta[clip3i(qp + beta_off, 0, 64)] << 2;
    return l;
}

/* sample accessor along the filtering direction: s(i) for i >= 0 is q_i, for i < 0 is p_{-i-1} */
#define S(i) ((int)pix[(i) * step])

static int dbf_dp(const uint16_t *pix, int step) { return abs(S(-3) - 2 * S(-2) + S(-1)); }
static int dbf_dq(const uint16_t *pix, int step) { return abs(S(0) - 2 * S(1) + S(2)); }

/* use_strong_filter_l0, rcn_df.c:77-123 */
static int dbf_strong_large(const uint16_t *pix, int step, int beta, int tc, int lp, int lq)
{
    int sp3 = abs(S(-4) - S(-1)), sq3 = abs(S(3) - S(0));
    if (lp == 7)      { sp3 += abs((S(-5) - S(-6)) - S(-7) + S(-8)); sp3 += abs(S(-4) - S(-8)) + 1; sp3 >>= 1; }
    else if (lp == 5) { sp3 += abs(S(-4) - S(-6)) + 1; sp3 >>= 1; }
    if (lq == 7)      { sq3 += abs((S(4) - S(5)) - S(6) + S(7)); sq3 += abs(S(7) - S(3)) + 1; sq3 >>= 1; }
    else if (lq == 5) { sq3 += abs(S(5) - S(3)) + 1; sq3 >>= 1; }
    return ((sp3 + sq3) < (beta * 3 >> 5)) && (abs(S(-1) - S(0)) < ((tc * 5 + 1) >> 1));
}

/* use_strong_filter_l1, rcn_df.c:125-139 */
static int dbf_strong_small(const uint16_t *pix, int step, int beta, int tc)
{
    return ((abs(S(-4) - S(-1)) + abs(S(3) - S(0))) < (beta >> 3)) && (abs(S(-1) - S(0)) < ((tc * 5 + 1) >> 1));
}

/* one line of the long ("large block") luma filters filter_{h,v}_{3,5,7}_{3,5,7}, rcn_df.c:217-846 */
static void dbf_long_line(uint16_t *pix, int step, int tc, int lp, int lq)
{
    static const int8_t f7[7] = { 59, 50, 41, 32, 23, 14, 5 }, f5[5] = { 58, 45, 32, 19, 6 }, f3[3] = { 53, 32, 11 };
    static const int8_t t7[7] = { 6, 5, 4, 3, 2, 1, 1 }, t3[3] = { 6, 4, 2 };
    int p[8], q[8];
    for (int i = 0; i < 8; ++i) { p[i] = S(-1 - i); q[i] = S(i); }
    int ref_p = (p[lp - 1] + p[lp] + 1) >> 1, ref_q = (q[lq - 1] + q[lq] + 1) >> 1, mid;
    if (lp == lq && lp == 7)
        mid = (2 * (p[0] + q[0]) + p[1] + p[2] + p[3] + p[4] + p[5] + p[6] + q[1] + q[2] + q[3] + q[4] + q[5] + q[6] + 8) >> 4;
    else if (lp == lq)                      /* 5,5 */
        mid = (2 * (p[0] + p[1] + p[2] + q[0] + q[1] + q[2]) + p[3] + p[4] + q[3] + q[4] + 8) >> 4;
    else if (lp + lq == 12)                 /* 7,5 / 5,7 */
        mid = (2 * (p[0] + p[1] + q[0] + q[1]) + p[2] + p[3] + p[4] + p[5] + q[2] + q[3] + q[4] + q[5] + 8) >> 4;
    else if (lp + lq == 8)                  /* 5,3 / 3,5 */
        mid = (p[0] + p[1] + p[2] + p[3] + q[0] + q[1] + q[2] + q[3] + 4) >> 3;
    else if (lp == 7)                       /* 7,3 */
        mid = (2 * (p[0] + q[0]) + p[1] + p[2] + p[3] + p[4] + p[5] + p[6] + q[0] + 3 * q[1] + 2 * q[2] + 8) >> 4;
    else                                    /* 3,7 */
        mid = (2 * (p[0] + q[0]) + q[1] + q[2] + q[3] + q[4] + q[5] + q[6] + p[0] + 3 * p[1] + 2 * p[2] + 8) >> 4;
    const int8_t *fp = lp == 7 ? f7 : lp == 5 ? f5 : f3, *fq = lq == 7 ? f7 : lq == 5 ? f5 : f3;
    const int8_t *tp = lp == 3 ? t3 : t7, *tq = lq == 3 ? t3 : t7;
    for (int i = 0; i < lp; ++i) {
        int cv = (tc * tp[i]) >> 1;
        pix[(-1 - i) * step] = (uint16_t)clip3i((mid * fp[i] + ref_p * (64 - fp[i]) + 32) >> 6, p[i] - cv, p[i] + cv);
    }
    for (int i = 0; i < lq; ++i) {
        int cv = (tc * tq[i]) >> 1;
        pix[i * step] = (uint16_t)clip3i((mid * fq[i] + ref_q * (64 - fq[i]) + 32) >> 6, q[i] - cv, q[i] + cv);
    }
}

/* filter_luma_strong_small_{h,v}, rcn_df.c:849-898 */
static void dbf_strong_line(uint16_t *pix, int step, int tc)
{
    const int p3 = S(-4), p2 = S(-3), p1 = S(-2), p0 = S(-1), q0 = S(0), q1 = S(1), q2 = S(2), q3 = S(3);
    pix[-3 * step] = (uint16_t)clip3i((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3, p2 - tc, p2 + tc);
    pix[-2 * step] = (uint16_t)clip3i((p2 + p1 + p0 + q0 + 2) >> 2, p1 - 2 * tc, p1 + 2 * tc);
    pix[-1 * step] = (uint16_t)clip3i((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3, p0 - 3 * tc, p0 + 3 * tc);
    pix[0]         = (uint16_t)clip3i((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3, q0 - 3 * tc, q0 + 3 * tc);
    pix[1 * step]  = (uint16_t)clip3i((p0 + q0 + q1 + q2 + 2) >> 2, q1 - 2 * tc, q1 + 2 * tc);
    pix[2 * step]  = (uint16_t)clip3i((p0 + q0 + q1 + 3 * q2 + 2 * q3 + 4) >> 3, q2 - tc, q2 + tc);
}

/* filter_luma_weak_{h,v}, rcn_df.c:900-958 */
static void dbf_weak_line(uint16_t *pix, int step, int tc, int ext_p, int ext_q)
{
    const int p2 = S(-3), p1 = S(-2), p0 = S(-1), q0 = S(0), q1 = S(1), q2 = S(2);
    const int tc2p = ext_p ? tc >> 1 : 0, tc2q = ext_q ? tc >> 1 : 0;
    int delta = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
    if (abs(delta) < tc * 10) {
        delta = clip3i(delta, -tc, tc);
        const int d1 = clip3i((((p2 + p0 + 1) >> 1) - p1 + delta) >> 1, -tc2p, tc2p);
        const int d2 = clip3i((((q2 + q0 + 1) >> 1) - q1 - delta) >> 1, -tc2q, tc2q);
        pix[-2 * step] = (uint16_t)clip_bd(p1 + d1);
        pix[-1 * step] = (uint16_t)clip_bd(p0 + delta);
        pix[0]         = (uint16_t)clip_bd(q0 - delta);
        pix[1 * step]  = (uint16_t)clip_bd(q1 + d2);
    }
}

/* filter_vertical_edge / filter_horizontal_edge, rcn_df.c:1433-1510, :2008-2085: one 4-line segment.
 * step = distance between samples across the edge, lstep = distance between the 4 lines. */
static void dbf_luma_segment(uint16_t *pix0, int step, int lstep, dbf_lim lim, int lp, int lq)
{
    const uint16_t *pix = pix0;
    const int dp0 = dbf_dp(pix, step), dq0 = dbf_dq(pix, step);
    pix = pix0 + 3 * lstep;
    const int dp3 = dbf_dp(pix, step), dq3 = dbf_dq(pix, step);
    const int d0 = dp0 + dq0, d3 = dp3 + dq3, beta = lim.beta, tc = lim.tc;
    if (d0 + d3 >= beta) return;
    int strong_large = 0;
    if (lp > 3 || lq > 3) {
        int dp0L = dp0, dq0L = dq0, dp3L = dp3, dq3L = dq3;
        if (lp > 3) {
            dp0L = (dp0L + dbf_dp(pix0 - 3 * step, step) + 1) >> 1;
            dp3L = (dp3L + dbf_dp(pix0 + 3 * lstep - 3 * step, step) + 1) >> 1;
        }
        if (lq > 3) {
            dq0L = (dq0L + dbf_dq(pix0 + 3 * step, step) + 1) >> 1;
            dq3L = (dq3L + dbf_dq(pix0 + 3 * lstep + 3 * step, step) + 1) >> 1;
        }
        const int d0L = dp0L + dq0L, d3L = dp3L + dq3L;
        strong_large = (d0L + d3L < beta) && (d0L < ((beta + 0x10) >> 5)) && (d3L < ((beta + 0x10) >> 5))
            && dbf_strong_large(pix0, step, beta, tc, lp, lq) && dbf_strong_large(pix0 + 3 * lstep, step, beta, tc, lp, lq);
    }
    if (strong_large) {
        for (int l = 0; l < 4; ++l) dbf_long_line(pix0 + l * lstep, step, tc, lp, lq);
        return;
    }
    int sw = lp > 2 && (d0 < ((beta + 4) >> 3)) && (d3 < ((beta + 4) >> 3))
        && dbf_strong_small(pix0, step, beta, tc) && dbf_strong_small(pix0 + 3 * lstep, step, beta, tc);
    if (sw) {
        for (int l = 0; l < 4; ++l) dbf_strong_line(pix0 + l * lstep, step, tc);
    } else {
        const int side = (beta + (beta >> 1)) >> 3;
        const int ext_p = (dp0 + dp3) < side && lp > 1;
        const int ext_q = (dq0 + dq3) < side && lp > 1;     /* sic: the reference tests max_l_p for Q too (rcn_df.c:1505) */
        for (int l = 0; l < 4; ++l) dbf_weak_line(pix0 + l * lstep, step, tc, ext_p, ext_q);
    }
}

/* filter_veritcal_edge_c / filter_horizontal_edge_c, rcn_df.c:1107-1148, :1279-1319: one 2-line chroma segment */
static void dbf_chroma_segment(uint16_t *pix0, int step, int lstep, dbf_lim lim, int large, int ctb_b)
{
    const int tc = lim.tc, beta = lim.beta;
    if (tc == 0 || beta == 0) return;
    int strong = 0;
    if (large) {
        int d[2];
        int ok = 1;
        for (int l = 0; l < 2; ++l) {
            const uint16_t *pix = pix0 + l * lstep;
            const int p3 = ctb_b ? S(-2) : S(-4);     /* src[(-stride*4) >> is_ctb_b] */
            const int dp = abs((ctb_b ? S(-2) : S(-3)) - 2 * S(-2) + S(-1));
            d[l] = dp + dbf_dq(pix, step);
            ok = ok && ((abs(p3 - S(-1)) + abs(S(3) - S(0))) < (beta >> 3)) && (abs(S(-1) - S(0)) < ((tc * 5 + 1) >> 1));
        }
        strong = ok && (d[0] + d[1] < beta) && (2 * d[0] < (beta >> 2)) && (2 * d[1] < (beta >> 2));
    }
    for (int l = 0; l < 2; ++l) {
        uint16_t *pix = pix0 + l * lstep;
        const int p3 = S(-4), p2 = S(-3), p1 = S(-2), p0 = S(-1), q0 = S(0), q1 = S(1), q2 = S(2), q3 = S(3);
        if (strong) {
            if (ctb_b) {
                pix[-1 * step] = (uint16_t)clip3i((3 * p1 + 2 * p0 + q0 + q1 + q2 + 4) >> 3, p0 - tc, p0 + tc);
                pix[0]         = (uint16_t)clip3i((2 * p1 + p0 + 2 * q0 + q1 + q2 + q3 + 4) >> 3, q0 - tc, q0 + tc);
                pix[1 * step]  = (uint16_t)clip3i((p1 + p0 + q0 + 2 * q1 + q2 + 2 * q3 + 4) >> 3, q1 - tc, q1 + tc);
                pix[2 * step]  = (uint16_t)clip3i((p0 + q0 + q1 + 2 * q2 + 3 * q3 + 4) >> 3, q2 - tc, q2 + tc);
            } else {
                pix[-3 * step] = (uint16_t)clip3i((3 * p3 + 2 * p2 + p1 + p0 + q0 + 4) >> 3, p2 - tc, p2 + tc);
                pix[-2 * step] = (uint16_t)clip3i((2 * p3 + p2 + 2 * p1 + p0 + q0 + q1 + 4) >> 3, p1 - tc, p1 + tc);
                pix[-1 * step] = (uint16_t)clip3i((p3 + p2 + p1 + 2 * p0 + q0 + q1 + q2 + 4) >> 3, p0 - tc, p0 + tc);
                pix[0]         = (uint16_t)clip3i((p2 + p1 + p0 + 2 * q0 + q1 + q2 + q3 + 4) >> 3, q0 - tc, q0 + tc);
                pix[1 * step]  = (uint16_t)clip3i((p1 + p0 + q0 + 2 * q1 + q2 + 2 * q3 + 4) >> 3, q1 - tc, q1 + tc);
                pix[2 * step]  = (uint16_t)clip3i((p0 + q0 + q1 + 2 * q2 + 3 * q3 + 4) >> 3, q2 - tc, q2 + tc);
            }
        } else {
            const int delta = clip3i(((q0 << 2) - (p0 << 2) + p1 - q1 + 4) >> 3, -tc, tc);
            pix[-1 * step] = (uint16_t)clip_bd(p0 + delta);
            pix[0]         = (uint16_t)clip_bd(q0 - delta);
        }
    }
}
#undef S

/* rcn_dbf_ctu over the whole picture: every vertical edge, then every horizontal edge (rcn_df.c:2169-2198) */
void oracle_dbf(const oracle_pic *pic, const ovhip_dbf_planes *pl)
{
    const int w4 = pl->w4, h4 = pl->h4, w4c = (w4 + 1) >> 1;
    for (int dir = 0; dir < 2; ++dir) {
        /* luma */
        const uint16_t *lw = dir ? pl->luma_h : pl->luma_v;
        for (int uy = 0; uy < h4; ++uy) {
            for (int ux = 0; ux < w4; ++ux) {
                const int v = lw[uy * w4 + ux];
                if (!(v & 3)) continue;
                dbf_lim lim = dbf_limits(v >> 8, v & 3, pl->tc_offset, pl->beta_offset);
                if (!(lim.tc || lim.beta)) continue;
                uint16_t *p = pic->y + (uy * 4) * pic->stride_y + ux * 4;
                dbf_luma_segment(p, dir ? pic->stride_y : 1, dir ? 1 : pic->stride_y, lim, (v >> 2) & 7, (v >> 5) & 7);
            }
        }
        /* chroma */
        for (int comp = 0; comp < 2; ++comp) {
            const uint16_t *cw = dir ? (comp ? pl->cr_h : pl->cb_h) : (comp ? pl->cr_v : pl->cb_v);
            uint16_t *plane = comp ? pic->cr : pic->cb;
            for (int uy = 0; uy < h4; uy += dir ? 2 : 1) {
                for (int ux = 0; ux < w4; ux += dir ? 1 : 2) {
                    const int v = dir ? cw[(uy >> 1) * w4 + ux] : cw[uy * w4c + (ux >> 1)];
                    if (!(v & OVHIP_DBF_C_ON)) continue;
                    dbf_lim lim = dbf_limits(v >> 8, 1 + !!(v & OVHIP_DBF_C_BS2), pl->tc_offset, pl->beta_offset);
                    uint16_t *p = plane + (uy * 2) * pic->stride_c + ux * 2;
                    dbf_chroma_segment(p, dir ? pic->stride_c : 1, dir ? 1 : pic->stride_c, lim,
                                       !!(v & OVHIP_DBF_C_LARGE), !!(v & OVHIP_DBF_C_CTB_B));
                }
            }
        }
    }
}

/* ====================================================================================
 * K13: sample adaptive offset
 * ================================================================================== */
/* sao_band_filter / sao_edge_filter / rcn_sao_ctu, rcn_sao.c:46-188.  Picture-level restatement:
 * every sample uses the parameters of the CTU that contains it; samples on the picture border
 * whose class neighbour lies outside are left unfiltered (the is_border row/column skips).  Borders of a rect entry (tile) inside
 * the picture count like the picture's: is_border comes from the ENTRY-local CTU index (rcn_sao.c:211-214, :253-257), carried
 * per CTU in ovhip_sao_ctu.border. */
void oracle_sao(const oracle_pic *dst, const oracle_pic *src, const ovhip_sao_ctu *prm, int log2_ctu)
{
    static const int8_t pos[4][2][2] = { { { -1, 0 }, { 1, 0 } }, { { 0, -1 }, { 0, 1 } }, { { -1, -1 }, { 1, 1 } }, { { 1, -1 }, { -1, 1 } } };
    const int nb_ctu_w = (src->w + (1 << log2_ctu) - 1) >> log2_ctu;
    /* quirk kept: with a single CTU row rcn_sao_first_pix_rows() passes OV_BOUNDARY_BOTTOM_RECT for its
     * 6-row band (rcn_sao.c:262), so the band's last row is skipped for non-horizontal edge classes */
    const int one_row = src->h <= (1 << log2_ctu);
    for (int c = 0; c < 3; ++c) {
        const int sh = c ? 1 : 0, w = src->w >> sh, h = src->h >> sh, l2 = log2_ctu - sh;
        int ss, ds;
        const uint16_t *s = plane_ptr(src, c, &ss);
        uint16_t *d = plane_ptr(dst, c, &ds);
        for (int y = 0; y < h; ++y) {
            for (int x = 0; x < w; ++x) {
                const ovhip_sao_ctu *p = &prm[(y >> l2) * nb_ctu_w + (x >> l2)];
                const int v = s[y * ss + x];
                int o = v;
                if (p->type[c] == OVHIP_SAO_BAND) {
                    const int k = ((v >> (BD - 5)) - p->band_position[c]) & 31;
                    if (k < 4) o = clip_bd(v + p->offset_val[c][k]);
                } else if (p->type[c] == OVHIP_SAO_EDGE) {
                    const int eo = p->eo_class[c];
                    const int b = p->border, cs = 1 << l2;
                    const int cx0 = x & ~(cs - 1), cy0 = y & ~(cs - 1);
                    const int cx1 = (cx0 + cs < w ? cx0 + cs : w) - 1, cy1 = (cy0 + cs < h ? cy0 + cs : h) - 1;
                    const int skip = (eo != 1 && (x == 0 || x == w - 1)) || (eo != 0 && (y == 0 || y == h - 1))
                                     || (one_row && eo != 0 && y == (6 >> sh) - 1)
                                     || (eo != 1 && (((b & OVHIP_BORDER_LEFT) && x == cx0) || ((b & OVHIP_BORDER_RIGHT) && x == cx1)))
                                     || (eo != 0 && (((b & OVHIP_BORDER_UPPER) && y == cy0) || ((b & OVHIP_BORDER_BOTTOM) && y == cy1)
                                                     || ((b & OVHIP_BORDER_ONE_ROW) && y == cy0 + (6 >> sh) - 1)));
                    if (!skip) {
                        const int a = s[(y + pos[eo][0][1]) * ss + x + pos[eo][0][0]];
                        const int b = s[(y + pos[eo][1][1]) * ss + x + pos[eo][1][0]];
                        const int idx = 2 + (v > a) - (v < a) + (v > b) - (v < b);
                        o = clip_bd(v + p->offset_val[c][idx]);
                    }
                }
                d[y * ds + x] = (uint16_t)o;
            }
        }
    }
}

/* ====================================================================================
 * K14: adaptive loop filter (classification, luma 7x7, chroma 5x5) + CC-ALF
 * ================================================================================== */
typedef struct oracle_alf {       /* host twin of ovhip_alf_pic */
    const ovhip_alf_ctu *ctus;
    const int16_t *luma_coeff, *luma_clip, *chroma_coeff, *chroma_clip, *cc_coeff;
    uint8_t *class_scratch;
    int32_t log2_ctu_s;
} oracle_alf;

/* The rectangle the filter windows of a CTU's samples are clamped to: the picture, cut at those sides of the CTU that are borders
 * of its rect entry (tile) -- rcn_extend_filter_region pads there as it does at the picture border (rcn_ctu.c:361-508 with the
 * entry-local is_border of rcn_alf.c:1313-1318).  A window never reaches further than the neighbouring CTU. */
typedef struct { int x0, y0, x1, y1; } alf_rect;
static alf_rect alf_clamp_rect(int border, int cx0, int cy0, int ctu, int w, int h)
{
    alf_rect r = { 0, 0, w - 1, h - 1 };
    if (border & OVHIP_BORDER_LEFT) r.x0 = cx0;
    if (border & OVHIP_BORDER_UPPER) r.y0 = cy0;
    if ((border & OVHIP_BORDER_RIGHT) && cx0 + ctu - 1 < r.x1) r.x1 = cx0 + ctu - 1;
    if ((border & OVHIP_BORDER_BOTTOM) && cy0 + ctu - 1 < r.y1) r.y1 = cy0 + ctu - 1;
    return r;
}
static inline int pxr(const uint16_t *p, int stride, const alf_rect *r, int x, int y)
{
    return p[clip3i(y, r->y0, r->y1) * stride + clip3i(x, r->x0, r->x1)];
}

/* alf_derive_filter_idx, rcn_alf.c:283-345 */
static void alf_filter_idx(uint32_t sum_h, uint32_t sum_v, uint32_t sum_d, uint32_t sum_b, int is_vb, int *cls, int *tr)
{
    static const int th[16] = { 0, 1, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3, 4 };
    static const uint8_t tr_lut[8] = { 0, 1, 0, 2, 2, 3, 1, 3 };
    const uint32_t scale = is_vb ? 96 : 64;
    int c = th[clip3i((int)(((sum_h + sum_v) * scale) >> (BD + 4)), 0, 15)];
    uint32_t max_hv, min_hv, max_db, min_db, max_dir, min_dir;
    int dir_hv, dir_db, main_dir, sec_dir;
    if (sum_v > sum_h) { max_hv = sum_v; min_hv = sum_h; dir_hv = 1; } else { max_hv = sum_h; min_hv = sum_v; dir_hv = 3; }
    if (sum_d > sum_b) { max_db = sum_d; min_db = sum_b; dir_db = 0; } else { max_db = sum_b; min_db = sum_d; dir_db = 2; }
    if (max_db * min_hv > max_hv * min_db) { max_dir = max_db; min_dir = min_db; main_dir = dir_db; sec_dir = dir_hv; }
    else { max_dir = max_hv; min_dir = min_hv; main_dir = dir_hv; sec_dir = dir_db; }
    if (max_dir * 2 > 9 * min_dir) c += (((main_dir & 1) << 1) + 2) * 5;
    else if (max_dir > 2 * min_dir) c += (((main_dir & 1) << 1) + 1) * 5;
    *cls = c;
    *tr = tr_lut[(main_dir << 1) + (sec_dir >> 1)];
}

/* Laplacians of one row pair (r, r+1) over the 8 columns bx-2..bx+5, sub-sampled on the (r+c) even
 * lattice; `above` / `below` are the rows used as vertical neighbours of r and r+1
 * (rcn_alf_classif_{vbnd,novbnd}, rcn_alf.c:347-704) */
static void alf_lap_pair(const uint16_t *p, int stride, const alf_rect *rc, int bx, int r, int above, int below, uint32_t s[4])
{
    for (int k = 0; k < 4; ++k) {
        int c0 = bx - 2 + 2 * k, c1 = c0 + 1;
        int y1 = pxr(p, stride, rc, c0, r) << 1, y2 = pxr(p, stride, rc, c1, r + 1) << 1;
        s[0] += abs(y1 - pxr(p, stride, rc, c0, above) - pxr(p, stride, rc, c0, r + 1))            /* V */
              + abs(y2 - pxr(p, stride, rc, c1, r) - pxr(p, stride, rc, c1, below));
        s[1] += abs(y1 - pxr(p, stride, rc, c0 + 1, r) - pxr(p, stride, rc, c0 - 1, r))            /* H */
              + abs(y2 - pxr(p, stride, rc, c1 + 1, r + 1) - pxr(p, stride, rc, c1 - 1, r + 1));
        s[2] += abs(y1 - pxr(p, stride, rc, c0 - 1, above) - pxr(p, stride, rc, c0 + 1, r + 1))    /* D0 */
              + abs(y2 - pxr(p, stride, rc, c1 - 1, r) - pxr(p, stride, rc, c1 + 1, below));
        s[3] += abs(y1 - pxr(p, stride, rc, c0 - 1, r + 1) - pxr(p, stride, rc, c0 + 1, above))    /* D1 */
              + abs(y2 - pxr(p, stride, rc, c1 - 1, below) - pxr(p, stride, rc, c1 + 1, r));
    }
}

/* vb: virtual-boundary row in CTU-LOCAL luma rows as the reference derives it -- ctu_h - 4 for a
 * full-height CTU, pic_h for a truncated one (rcn_alf.c:722, :1346); it is compared with CTU-local
 * rows, so for truncated CTUs it only ever matches in pictures of a single CTU row. */
static void alf_classify_block(const oracle_pic *src, const alf_rect *rc, int bx, int by, int vb, int ctu_y0, int *cls, int *tr)
{
    const int lby = by - ctu_y0;
    uint32_t s[4] = { 0, 0, 0, 0 };
    int first = 0, last = 3, is_vb = 0;
    if (lby == vb - 4) { last = 2; is_vb = 1; }
    if (lby == vb)     { first = 1; is_vb = 1; }
    for (int k = first; k <= last; ++k) {
        const int r = by - 2 + 2 * k;
        int above = r - 1, below = r + 2;
        if (r - ctu_y0 + 2 == vb) below = r + 1;       /* pair (vb-2, vb-1): row vb is not available */
        if (r - ctu_y0 == vb)     above = r;           /* pair (vb, vb+1): row vb-1 is not available */
        alf_lap_pair(src->y, src->stride_y, rc, bx, r, above, below, s);
    }
    alf_filter_idx(s[1], s[0], s[2], s[3], is_vb, cls, tr);
}

static inline int alf_clipd(int clip, int ref, int a, int b)
{
    return clip3i(a - ref, -clip, clip) + clip3i(b - ref, -clip, clip);
}

/* rcn_alf_filter_line over the whole picture, rcn_alf.c:1285-1433 */
void oracle_alf_run(const oracle_pic *dst, const oracle_pic *src, const oracle_alf *a)
{
    const int ctu = 1 << a->log2_ctu_s, W = src->w, H = src->h;
    const int nb_ctu_w = (W + ctu - 1) / ctu;
    /* ---- luma ---- */
    for (int y = 0; y < H; ++y) {
        for (int x = 0; x < W; ++x) {
            const ovhip_alf_ctu *c = &a->ctus[(y / ctu) * nb_ctu_w + x / ctu];
            const int cur = src->y[y * src->stride_y + x];
            int out = cur;
            if (c->flags & 4) {
                const int ctu_y0 = (y / ctu) * ctu, truncated = ctu_y0 + ctu > H;
                const int vb = truncated ? H : ctu - 4;
                const int ctu_h = truncated ? H - ctu_y0 : ctu;
                const int last_local = (ctu_y0 + ctu_h - 1) & (ctu - 1);
                /* check_virtual_bound, rcn_alf.c:1274-1283: selects alf.luma[1] (VB variant) */
                const int req_vb = (last_local < vb && last_local >= vb - 4) || (last_local >= vb && last_local <= vb + 3);
                int cls, tr;
                const alf_rect rc = alf_clamp_rect(c->border, (x / ctu) * ctu, ctu_y0, ctu, W, H);
                alf_classify_block(src, &rc, x & ~3, y & ~3, vb, ctu_y0, &cls, &tr);
                const int16_t *f = a->luma_coeff + c->luma_set * OVHIP_ALF_LUMA_SET_SIZE + tr * 25 * 13 + cls * 13;
                const int16_t *cl = a->luma_clip + c->luma_set * OVHIP_ALF_LUMA_SET_SIZE + tr * 25 * 13 + cls * 13;
                int d = 3, near = 0;
                if (req_vb) {
                    const int ly = y - ctu_y0;
                    if (ly < vb && ly >= vb - 4) d = vb - 1 - ly;
                    else if (ly >= vb && ly <= vb + 3) d = ly - vb;
                    near = (ly == vb - 1) || (ly == vb);
                }
                const int o1 = d < 1 ? d : 1, o2 = d < 2 ? d : 2, o3 = d < 3 ? d : 3;
                const uint16_t *p = src->y; const int st = src->stride_y;
#define L(dx, dy) pxr(p, st, &rc, x + (dx), y + (dy))
                int sum = 0;
                sum += f[0] * alf_clipd(cl[0], cur, L(0, o3), L(0, -o3));
                sum += f[1] * alf_clipd(cl[1], cur, L(1, o2), L(-1, -o2));
                sum += f[2] * alf_clipd(cl[2], cur, L(0, o2), L(0, -o2));
                sum += f[3] * alf_clipd(cl[3], cur, L(-1, o2), L(1, -o2));
                sum += f[4] * alf_clipd(cl[4], cur, L(2, o1), L(-2, -o1));
                sum += f[5] * alf_clipd(cl[5], cur, L(1, o1), L(-1, -o1));
                sum += f[6] * alf_clipd(cl[6], cur, L(0, o1), L(0, -o1));
                sum += f[7] * alf_clipd(cl[7], cur, L(-1, o1), L(1, -o1));
                sum += f[8] * alf_clipd(cl[8], cur, L(-2, o1), L(2, -o1));
                sum += f[9] * alf_clipd(cl[9], cur, L(3, 0), L(-3, 0));
                sum += f[10] * alf_clipd(cl[10], cur, L(2, 0), L(-2, 0));
                sum += f[11] * alf_clipd(cl[11], cur, L(1, 0), L(-1, 0));
#undef L
                sum = near ? (sum + 512) >> 10 : (sum + 64) >> 7;
                out = clip_bd(sum + cur);
            }
            dst->y[y * dst->stride_y + x] = (uint16_t)out;
        }
    }
    /* ---- chroma + CC-ALF ---- */
    const int Wc = W / 2, Hc = H / 2, ctuc = ctu / 2;
    for (int comp = 1; comp < 3; ++comp) {
        const uint16_t *p = comp == 1 ? src->cb : src->cr;
        uint16_t *o = comp == 1 ? dst->cb : dst->cr;
        const int st = src->stride_c;
        for (int y = 0; y < Hc; ++y) {
            for (int x = 0; x < Wc; ++x) {
                const ovhip_alf_ctu *c = &a->ctus[(y / ctuc) * nb_ctu_w + x / ctuc];
                const int ctu_y0 = (y / ctuc) * ctu;                /* luma row of the CTU */
                const int truncated = ctu_y0 + ctu > H;
                const int cur = p[y * st + x];
                int out = cur;
                const alf_rect rcc = alf_clamp_rect(c->border, (x / ctuc) * ctuc, (y / ctuc) * ctuc, ctuc, Wc, Hc);
                if (c->flags & (comp == 1 ? 2 : 1)) {
                    const int alt = comp == 1 ? c->cb_alt : c->cr_alt;
                    const int16_t *f = a->chroma_coeff + alt * 7, *cl = a->chroma_clip + alt * 7;
                    const int vb = truncated ? H / 2 : (ctu - 4) / 2;
                    const int ly = y & (ctuc - 1);
                    int d = 2;
                    if (ly < vb && ly >= vb - 2) d = vb - 1 - ly;
                    else if (ly >= vb && ly <= vb + 1) d = ly - vb;
                    const int near = (ly == vb - 1) || (ly == vb);
                    const int o1 = d < 1 ? d : 1, o2 = d < 2 ? d : 2;
#define Cc(dx, dy) pxr(p, st, &rcc, x + (dx), y + (dy))
                    int sum = 0;
                    sum += f[0] * alf_clipd(cl[0], cur, Cc(0, o2), Cc(0, -o2));
                    sum += f[1] * alf_clipd(cl[1], cur, Cc(1, o1), Cc(-1, -o1));
                    sum += f[2] * alf_clipd(cl[2], cur, Cc(0, o1), Cc(0, -o1));
                    sum += f[3] * alf_clipd(cl[3], cur, Cc(-1, o1), Cc(1, -o1));
                    sum += f[4] * alf_clipd(cl[4], cur, Cc(2, 0), Cc(-2, 0));
                    sum += f[5] * alf_clipd(cl[5], cur, Cc(1, 0), Cc(-1, 0));
#undef Cc
                    sum = near ? (sum + 512) >> 10 : (sum + 64) >> 7;
                    out = clip_bd(sum + cur);
                }
                const int cc = comp == 1 ? c->cc_cb_idx : c->cc_cr_idx;
                if (cc) {
                    /* cc_alf_filterBlk, rcn_alf.c:740-804; vbPos is NOT divided by the chroma scale in the
                     * non-truncated case (rcn_alf.c:1417) and is compared with the CTU-local LUMA row */
                    const int16_t *f = a->cc_coeff + ((comp - 1) * 4 + (cc - 1)) * 8;
                    const int vbpos = truncated ? H / 2 : ctu - 4;
                    const int pos = (y << 1) & (ctu - 1);
                    int r1 = 1, r2 = -1, r3 = 2;
                    if (pos == vbpos - 2 || pos == vbpos + 1) r3 = r1;
                    else if (pos == vbpos - 1 || pos == vbpos) r1 = r2 = r3 = 0;
                    const int lx = x << 1, lyy = y << 1;
                    const alf_rect rcl = alf_clamp_rect(c->border, (x / ctuc) * ctu, ctu_y0, ctu, W, H);
#define Ly(dx, dy) pxr(src->y, src->stride_y, &rcl, lx + (dx), lyy + (dy))
                    const int cy = Ly(0, 0);
                    int sum = 0;
                    sum += f[0] * (Ly(0, r2) - cy);
                    sum += f[1] * (Ly(-1, 0) - cy);
                    sum += f[2] * (Ly(1, 0) - cy);
                    sum += f[3] * (Ly(-1, r1) - cy);
                    sum += f[4] * (Ly(0, r1) - cy);
                    sum += f[5] * (Ly(1, r1) - cy);
                    sum += f[6] * (Ly(0, r3) - cy);
#undef Ly
                    sum = (sum + 64) >> 7;
                    sum = clip_bd(sum + (1 << BD >> 1));
                    sum += out - (1 << BD >> 1);
                    out = clip_bd(sum);
                }
                o[y * dst->stride_c + x] = (uint16_t)out;
            }
        }
    }
}

/* ====================================================================================
 * TMVP motion plane: where the refined vectors go
 * ================================================================================== */
/* The sequential form of what the reference does per CTU: the caller overwrites entries of the CTU-local array
 * tmvp_mv[l].mvs[16 * 16] with what rcn_dmvr_mv_refine returned (vcl_coding_unit.c:2629-2645: entry
 * ((x0 + 7 + j * 16) >> 3) + ((y0 + 7 + i * 16) >> 3) * 16, + 1 for 16-wide, + 16 for 16-high blocks, + 17 for both), then
 * tmvp_store_mv copies the first nb_tmvp_unit entries of the first nb_tmvp_unit rows of that array into the picture's plane
 * at ctb_offset + i * pln_stride (drv_lines.c:288-310).  units: the refined units (<= 16x16 each, picture coordinates);
 * out: 4 entries per unit as ovhip_tmvp_cells_launch writes them. */
void oracle_tmvp_cells(const ovhip_mc_unit *units, uint32_t n, const int32_t *refined, int log2_ctu, int nb_ctb_w, ovhip_tmvp_cell *out)
{
    const int nb_tmvp_unit = (1 << log2_ctu) >> 3, pln_stride = nb_tmvp_unit * nb_ctb_w;
    for (uint32_t u = 0; u < n; ++u) {
        const ovhip_mc_unit *t = &units[u];
        for (int k = 0; k < 4; ++k) { out[4 * u + k].cell = OVHIP_TMVP_NONE; out[4 * u + k].mv0x = out[4 * u + k].mv0y = out[4 * u + k].mv1x = out[4 * u + k].mv1y = 0; }
        if (!(t->flags & OVHIP_MC_DMVR)) continue;
        const int ctb_x = t->x >> log2_ctu, ctb_y = t->y >> log2_ctu;
        const int x0 = t->x - (ctb_x << log2_ctu), y0 = t->y - (ctb_y << log2_ctu);
        const int log2_w = t->w > 8 ? 4 : 3, log2_h = t->h > 8 ? 4 : 3;
        const int base = ((x0 + 7) >> 3) + ((y0 + 7) >> 3) * 16;
        int idx[4], n_idx = 0;
        idx[n_idx++] = base;
        if (log2_w > 3) idx[n_idx++] = base + 1;
        if (log2_h > 3) { idx[n_idx++] = base + 16; if (log2_w > 3) idx[n_idx++] = base + 16 + 1; }
        const int ctb_offset = (ctb_x + ctb_y * pln_stride) * nb_tmvp_unit;
        for (int q = 0; q < n_idx; ++q) {
            const int row = idx[q] >> 4, col = idx[q] & 15;           /* entry of the 16-stride array */
            if (row >= nb_tmvp_unit || col >= nb_tmvp_unit) continue; /* not among what tmvp_store_mv copies */
            /* slot: 0 base, 1 right, 2 below, 3 below-right (the order of ovhip_tmvp_cells_launch) */
            const int slot = (idx[q] - base == 1) ? 1 : (idx[q] - base == 16 ? 2 : (idx[q] - base == 17 ? 3 : 0));
            ovhip_tmvp_cell *c = &out[4 * u + slot];
            c->cell = (uint32_t)(ctb_offset + row * pln_stride + col);
            c->mv0x = refined[4 * u]; c->mv0y = refined[4 * u + 1]; c->mv1x = refined[4 * u + 2]; c->mv1y = refined[4 * u + 3];
        }
    }
}

/* ====================================================================================
 * Intra prediction and the ordered pass
 * ================================================================================== */
#include "ovvc_oracle_intra.c"
