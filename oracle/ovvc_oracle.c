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
static void itx_one(const oracle_pic *pic, const ovhip_tb_cmd *c, const int16_t *arena)
{
    const int log2_w = c->log2_w, log2_h = c->log2_h;
    const int tb_w = 1 << log2_w, tb_h = 1 << log2_h;
    const int kind = c->kind & 0x7f, raster = !!(c->kind & OVHIP_TB_FLAG_RASTER);
    const int cw = tb_w > 32 ? 32 : tb_w, ch = tb_h > 32 ? 32 : tb_h;   /* stored coefficient extent */
    const int16_t *src = arena + c->coef_off;
    static _Thread_local int16_t coef[32 * 32], tmp[64 * 64], res[64 * 64];

    memset(coef, 0, sizeof(int16_t) * cw * ch);
    if (raster) {
        for (int i = 0; i < tb_w * tb_h; ++i)
            coef[i] = kind == OVHIP_TB_TS_RAW ? src[i] : dequant1(src[i], c->dq_scale, c->dq_shift, c->dq_neg);
    } else {
        uint64_t map = c->sig_sb_map;
        int n = 0;
        while (map) {
            int b = __builtin_ctzll(map);
            map &= map - 1;
            int16_t *d = coef + (b >> 3) * 4 * cw + (b & 7) * 4;
            for (int r = 0; r < 4; ++r)
                for (int q = 0; q < 4; ++q)
                    d[r * cw + q] = dequant1(src[n * 16 + r * 4 + q], c->dq_scale, c->dq_shift, c->dq_neg);
            ++n;
        }
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
        tr_pass(coef, tmp, cw, c->tr_v, log2_h, nb_row, 7);          /* TR_SHIFT_V */
        tr_pass(tmp, res, tb_h, c->tr_h, log2_w, tb_h, 20 - BD);     /* TR_SHIFT_H */
    }

    int stride;
    uint16_t *d = plane_ptr(pic, c->plane, &stride) + c->y * stride + c->x;
    residual_apply(d, stride, res, tb_w, tb_h, c->res_mode, c->c_scale);
    if (c->plane2 != 0xff) {
        d = plane_ptr(pic, c->plane2, &stride) + c->y * stride + c->x;
        residual_apply(d, stride, res, tb_w, tb_h, c->res_mode2, c->c_scale);
    }
}

void oracle_itx(const oracle_pic *pic, const ovhip_tb_cmd *cmds, uint32_t n, const int16_t *arena)
{
    for (uint32_t i = 0; i < n; ++i) itx_one(pic, &cmds[i], arena);
}

/* ====================================================================================
 * K5/K6/K11: motion compensation, uni / bi / BCW, luma + chroma, LMCS forward reshape
 * ================================================================================== */

/* 14-bit intermediate prediction of one block, exactly as put_vvc_{pel,qpel,epel}*_{h,v,hv}
 * compute it (rcn_mc.c:402-420, :903-985, :1188-1270).  The four reference variants are one
 * separable filter whose integer-position row is the identity tap (see vvc_mc_taps.h):
 *   t = F_h(src) >> (BD-8);  P = F_v(t) >> 6.
 * Samples outside the picture are replicated (emulate_block_border, rcn_inter.c:148-225). */
static void predict14(int16_t *out, int ow, const uint16_t *ref, int rstride, int rw, int rh,
                      int px, int py, int w, int h, const int8_t *fh, const int8_t *fv, int ntaps)
{
    const int before = ntaps == 8 ? 3 : 1;
    int32_t tmp[(16 + 7) * 16];
    for (int y = 0; y < h + ntaps - 1; ++y) {
        int sy = clip3i(py + y - before, 0, rh - 1);
        for (int x = 0; x < w; ++x) {
            int32_t s = 0;
            for (int t = 0; t < ntaps; ++t) {
                int sx = clip3i(px + x + t - before, 0, rw - 1);
                s += fh[t] * (int32_t)ref[sy * rstride + sx];
            }
            tmp[y * 16 + x] = (int16_t)(s >> (BD - 8));
        }
    }
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w; ++x) {
            int32_t s = 0;
            for (int t = 0; t < ntaps; ++t) s += fv[t] * tmp[(y + t) * 16 + x];
            out[y * ow + x] = (int16_t)(s >> 6);
        }
    }
}

static void mc_plane(const oracle_pic *dst, const oracle_pic *refs, const ovhip_mc_unit *u, int plane,
                     const uint16_t *lmcs_fwd)
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
        int rstride;
        const uint16_t *r = plane_ptr(rp, plane, &rstride);
        const int8_t *fh, *fv;
        if (!c) {
            int fx = mvx & 15, fy = mvy & 15;
            if (u->flags & OVHIP_MC_FILT_4x4) { fh = ovt_mc_luma4[fx]; fv = ovt_mc_luma4[fy]; }
            else {
                if (u->flags & OVHIP_MC_HPEL_FILT) { if (fx == 8) fx = 16; if (fy == 8) fy = 16; }
                fh = ovt_mc_luma[fx]; fv = ovt_mc_luma[fy];
            }
            predict14(p[l], 16, r, rstride, rp->w, rp->h, x + (mvx >> 4), y + (mvy >> 4), w, h, fh, fv, 8);
        } else {
            fh = ovt_mc_chroma[mvx & 31]; fv = ovt_mc_chroma[mvy & 31];
            predict14(p[l], 16, r, rstride, rp->w >> 1, rp->h >> 1, x + (mvx >> 5), y + (mvy >> 5), w, h, fh, fv, 4);
        }
    }
    for (int j = 0; j < h; ++j) {
        for (int i = 0; i < w; ++i) {
            int v;
            if (u->dir != 3) {
                v = clip_bd((p[u->dir - 1][j * 16 + i] + 8) >> 4);                 /* uni: rcn_mc.c:448-533 */
            } else if (u->w0 == 4 && u->w1 == 4) {
                v = clip_bd((p[0][j * 16 + i] + p[1][j * 16 + i] + 16) >> 5);      /* bi: rcn_mc.c:422-444, :987-1098 */
            } else {
                v = clip_bd((p[1][j * 16 + i] * u->w1 + p[0][j * 16 + i] * u->w0 + 64) >> 7); /* BCW: rcn_mc.c:1480-1610 */
            }
            if (!c && (u->flags & OVHIP_MC_LMCS) && lmcs_fwd) v = lmcs_fwd[v & PIX_MAX]; /* rcn_lmcs.c:275-295 */
            d[j * dstride + i] = (uint16_t)v;
        }
    }
}

/* rcn_mcp_l/_c, rcn_motion_compensation_b_l/_c (rcn_inter.c:520-602, :1391-1554, :1822-1904) */
void oracle_mc(const oracle_pic *dst, const oracle_pic *refs, uint32_t n_refs,
               const ovhip_mc_unit *units, uint32_t n, const uint16_t *lmcs_fwd)
{
    (void)n_refs;
    for (uint32_t i = 0; i < n; ++i) {
        const ovhip_mc_unit *u = &units[i];
        if (!(u->flags & OVHIP_MC_NO_LUMA)) mc_plane(dst, refs, u, 0, lmcs_fwd);
        if (!(u->flags & OVHIP_MC_NO_CHROMA)) { mc_plane(dst, refs, u, 1, lmcs_fwd); mc_plane(dst, refs, u, 2, lmcs_fwd); }
    }
}
