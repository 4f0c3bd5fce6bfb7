/* ovvc_record.c -- host-side recorder of the MI355X back-end (plain C, no GPU needed).
 *
 * The reference reconstructs each block inside the orchestrator slots of struct RCNFunctions,
 * called from the CABAC parse loop.  The GPU path defers execution: these
 * functions perform the *control* part of those orchestrators on the host -- everything that
 * depends on decoder state rather than on samples -- and emit self-contained fixed-size
 * commands for the device kernels.  Behaviour restated from (never copied):
 *   rcn_tu_st / rcn_tu_l / rcn_tu_c        libovvc/rcn_transform_tree.c:1228-1382
 *   rcn_residual / rcn_residual_c           libovvc/rcn_transform_tree.c:415-506, :553-628
 *   rcn_res_c / rcn_jcbcr                   libovvc/rcn_transform_tree.c:720-867
 *   transform-skip paths                    libovvc/rcn_transform_tree.c:672-716, :1208-1225
 *   derive_dequant_{sdh,dpq,ts}             libovvc/rcn_dequant.c:92-158
 *   drv_lfnst_mode_l / process_lfnst(_luma) libovvc/drv_lfnst.c:42-156
 *   ict.ict[][] selection                   libovvc/rcn_residuals.c:231-331
 *   rcn_mcp_b dispatch, clip_mv, identical motion, AMVR half-pel, BCW
 *                                           libovvc/rcn_inter.c:89-109, :256-268, :520-602, :2769-2813
 */
#include <stdlib.h>
#include <string.h>
#include "ovvc_hip.h"
#include "ovvc_record_priv.h"

/* CUFlags bits this path looks at (libovvc/cu_utils.h:44-60) */
#define CUF_PRED_MODE_INTRA   (1u << 1)
#define CUF_MIP               (1u << 2)
#define CUF_BDPCM_LUMA        (1u << 8)
#define CUF_BDPCM_CHROMA      (1u << 9)
#define CUF_BDPCM_LUMA_DIR    (1u << 10)
#define CUF_BDPCM_CHROMA_DIR  (1u << 11)

ovhip_recorder *
ovhip_rec_create_ex(int32_t pic_w, int32_t pic_h, const ovhip_allocator *a)
{
    if (pic_w <= 0 || pic_h <= 0 || (a && (!a->alloc || !a->free))) return NULL;
    ovhip_recorder *r = (ovhip_recorder *)calloc(1, sizeof(*r));
    if (!r) return NULL;
    r->pic_w = pic_w;
    r->pic_h = pic_h;
    r->dense_planes = 1;
    r->log2_ctu = 7;
    if (a) { r->al = *a; r->has_al = 1; }
    return r;
}

ovhip_recorder *ovhip_rec_create(int32_t pic_w, int32_t pic_h) { return ovhip_rec_create_ex(pic_w, pic_h, NULL); }

void
ovhip_rec_free_(ovhip_recorder *r, void *p)
{
    if (!p) return;
    if (r->has_al) r->al.free(r->al.user, p); else free(p);
}

void
ovhip_rec_destroy(ovhip_recorder *r)
{
    if (!r) return;
    void *bufs[] = { r->tb, r->coef, r->mc, r->mcx, r->aff, r->aff_side, r->reg, r->tb_split, r->ciip, r->edge_v, r->edge_h,
                     r->itask, r->itask_sorted, r->itask_ctu, r->ictu };
    free(r->ilevel_start); free(r->ctu_count);
    ovhip_rec_intra_free_(r);
    for (size_t i = 0; i < sizeof(bufs) / sizeof(bufs[0]); ++i) ovhip_rec_free_(r, bufs[i]);
    ovhip_rec_dbf_free_(r);
    free(r);
}

void
ovhip_rec_reset(ovhip_recorder *r)
{
    if (!r) return;
    r->n_tb = r->n_coef = r->n_mc = r->n_mcx = r->n_aff = r->n_side = r->n_reg = r->n_ciip = 0;
    r->n_edge_v = r->n_edge_h = 0;
    r->n_dbf_off = 0;
    ovhip_rec_intra_reset_(r);
    ovhip_rec_dbf_reset_(r);
}

void ovhip_rec_set_dense_dbf_planes(ovhip_recorder *r, int on) { if (r) r->dense_planes = !!on; }

/* Capacity for a picture of the recorder's size up front, so that recording never has to grow an array: with a page-locked allocator a
 * growth is hipHostMalloc + copy + hipHostFree (0.1-1.4 ms each, measured on the live decoder: 113 of them, 37 ms, inside ONE timed
 * repetition of a 33-picture stream -- every frame thread meets its first I picture, its first P picture ... at some point).  Sized
 * from what 4K pictures of the seeded streams record (I pictures: 80 k ordered tasks, 8 MB of coefficients), with P = w x h:
 * 30 MB at 4K (the side arena of the affine units alone 5.5: the seeded streams are affine-heavy).  Arrays still grow past this if a picture needs it. */
int
ovhip_rec_reserve_for_picture(ovhip_recorder *r)
{
    if (!r) return OVHIP_EINVAL;
    const size_t P = (size_t)r->pic_w * r->pic_h;
    int bad = 0;
    bad |= ovhip_rec_grow_(r, (void **)&r->coef, &r->cap_coef, P / 2 + 1024, sizeof(int16_t));
    bad |= ovhip_rec_grow_(r, (void **)&r->tb, &r->cap_tb, P / 64 + 1024, sizeof(ovhip_tb_cmd));
    bad |= ovhip_rec_grow_(r, (void **)&r->tb_split, &r->cap_split, P / 64 + 1024, sizeof(ovhip_tb_cmd));
    bad |= ovhip_rec_grow_(r, (void **)&r->mc, &r->cap_mc, P / 128 + 1024, sizeof(ovhip_mc_unit));
    bad |= ovhip_rec_grow_(r, (void **)&r->mcx, &r->cap_mcx, P / 256 + 1024, sizeof(ovhip_mc_unit));
    bad |= ovhip_rec_grow_(r, (void **)&r->aff, &r->cap_aff, P / 1024 + 256, sizeof(ovhip_aff_unit));
    bad |= ovhip_rec_grow_(r, (void **)&r->aff_side, &r->cap_side, P / 6 + 1024, sizeof(int32_t));
    bad |= ovhip_rec_grow_(r, (void **)&r->reg, &r->cap_reg, P / 4096 + 64, sizeof(ovhip_lmcs_region));
    bad |= ovhip_rec_grow_(r, (void **)&r->edge_v, &r->cap_edge_v, P / 32 + 2048, sizeof(ovhip_dbf_edge));
    bad |= ovhip_rec_grow_(r, (void **)&r->edge_h, &r->cap_edge_h, P / 32 + 2048, sizeof(ovhip_dbf_edge));
    bad |= ovhip_rec_grow_(r, (void **)&r->itask, &r->cap_itask, P / 64 + 1024, sizeof(ovhip_itask));
    bad |= ovhip_rec_grow_(r, (void **)&r->itask_sorted, &r->cap_isorted, P / 64 + 1024, sizeof(ovhip_itask));
    return bad ? OVHIP_ENOMEM : OVHIP_OK;
}

#define ACCESSOR(type, name, arr, cnt) \
    const type *name(const ovhip_recorder *r, size_t *n) { if (!r || !n) return NULL; *n = r->cnt; return r->arr; }
ACCESSOR(ovhip_tb_cmd, ovhip_rec_tb_cmds, tb, n_tb)
ACCESSOR(int16_t, ovhip_rec_coefs, coef, n_coef)
ACCESSOR(ovhip_mc_unit, ovhip_rec_mc_units, mc, n_mc)
ACCESSOR(ovhip_mc_unit, ovhip_rec_mcx_units, mcx, n_mcx)
ACCESSOR(ovhip_aff_unit, ovhip_rec_aff_units, aff, n_aff)
ACCESSOR(int32_t, ovhip_rec_aff_side, aff_side, n_side)

/* lengths of the arrays a band of CTU rows is cut from (ovhip_job_band) */
void
ovhip_rec_counts(const ovhip_recorder *r, ovhip_band_counts *out)
{
    if (!out) return;
    memset(out, 0, sizeof(*out));
    if (!r) return;
    out->n_tb = (uint32_t)r->n_tb; out->n_coef = (uint32_t)r->n_coef; out->n_mc = (uint32_t)r->n_mc; out->n_mcx = (uint32_t)r->n_mcx;
    out->n_aff = (uint32_t)r->n_aff; out->n_side = (uint32_t)r->n_side; out->n_reg = (uint32_t)r->n_reg; out->n_itask = (uint32_t)r->n_itask;
    out->n_edge_v = (uint32_t)r->n_edge_v; out->n_edge_h = (uint32_t)r->n_edge_h;
}

/* Arrays grow geometrically; with a caller-supplied allocator (pinned host memory in the engine, so that the flush
 * is plain asynchronous DMA) growth is allocate + copy + free, since such memory cannot be realloc'ed. */
int
ovhip_rec_grow_(ovhip_recorder *r, void **p, size_t *cap, size_t need, size_t elem)
{
    if (need <= *cap) return 0;
    size_t nc = *cap ? *cap : 1024;
    while (nc < need) nc *= 2;
    void *q;
    if (r->has_al) {
        q = r->al.alloc(r->al.user, nc * elem);
        if (!q) return -1;
        if (*p) { memcpy(q, *p, *cap * elem); r->al.free(r->al.user, *p); }
    } else {
        q = realloc(*p, nc * elem);
        if (!q) return -1;
    }
    *p = q; *cap = nc;
    return 0;
}
/* (the capacity test inline: the call is the rare path -- it used to be a PLT call per appended element) */
#define grow(p, cap, need, elem) ((need) <= *(cap) ? 0 : ovhip_rec_grow_(r, p, cap, need, elem))

/* Bulk append of already-recorded commands (replaying a stored command stream: fixtures, benchmarks, a picture
 * recorded by another recorder).  Offsets inside the commands (coef_off, side_off, region indices) are taken as they
 * are, i.e. the stream must be appended to an empty recorder, arena first. */
int
ovhip_rec_append_raw(ovhip_recorder *r, int which, const void *data, size_t n)
{
    if (!r || (n && !data)) return OVHIP_EINVAL;
    void **p; size_t *cnt, *cap, elem;
    switch (which) {
    case OVHIP_REC_TB:     p = (void **)&r->tb;       cnt = &r->n_tb;     cap = &r->cap_tb;     elem = sizeof(ovhip_tb_cmd); break;
    case OVHIP_REC_COEF:   p = (void **)&r->coef;     cnt = &r->n_coef;   cap = &r->cap_coef;   elem = sizeof(int16_t); break;
    case OVHIP_REC_MC:     p = (void **)&r->mc;       cnt = &r->n_mc;     cap = &r->cap_mc;     elem = sizeof(ovhip_mc_unit); break;
    case OVHIP_REC_MCX:    p = (void **)&r->mcx;      cnt = &r->n_mcx;    cap = &r->cap_mcx;    elem = sizeof(ovhip_mc_unit); break;
    case OVHIP_REC_AFF:    p = (void **)&r->aff;      cnt = &r->n_aff;    cap = &r->cap_aff;    elem = sizeof(ovhip_aff_unit); break;
    case OVHIP_REC_SIDE:   p = (void **)&r->aff_side; cnt = &r->n_side;   cap = &r->cap_side;   elem = sizeof(int32_t); break;
    case OVHIP_REC_REGION: p = (void **)&r->reg;      cnt = &r->n_reg;    cap = &r->cap_reg;    elem = sizeof(ovhip_lmcs_region); break;
    case OVHIP_REC_CIIP:   p = (void **)&r->ciip;     cnt = &r->n_ciip;   cap = &r->cap_ciip;   elem = sizeof(ovhip_ciip_unit); break;
    case OVHIP_REC_EDGE_V: p = (void **)&r->edge_v;   cnt = &r->n_edge_v; cap = &r->cap_edge_v; elem = sizeof(ovhip_dbf_edge); break;
    case OVHIP_REC_EDGE_H: p = (void **)&r->edge_h;   cnt = &r->n_edge_h; cap = &r->cap_edge_h; elem = sizeof(ovhip_dbf_edge); break;
    case OVHIP_REC_ITASK:  p = (void **)&r->itask;    cnt = &r->n_itask;  cap = &r->cap_itask;  elem = sizeof(ovhip_itask); break;
    default: return OVHIP_EINVAL;
    }
    if (grow(p, cap, *cnt + n, elem)) return OVHIP_ENOMEM;
    if (n) memcpy((char *)*p + *cnt * elem, data, n * elem);
    *cnt += n;
    return OVHIP_OK;
}

/* The deblocking offsets of a replayed stream (ovhip_rec_dbf_ctu collects them itself). */
int
ovhip_rec_set_dbf_offsets(ovhip_recorder *r, const ovhip_dbf_offsets *o, int n)
{
    if (!r || !o || n < 0 || n > OVHIP_DBF_MAX_OFFSETS) return OVHIP_EINVAL;
    r->dbf_off = *o; r->n_dbf_off = n;
    return OVHIP_OK;
}

/* ---------------------------------------------------------------- dequantisation params */
struct dq { int16_t scale; uint8_t shift, neg; };

static const int16_t iq_scale[2][6] = { { 40, 45, 51, 57, 64, 72 }, { 57, 64, 72, 80, 90, 102 } };

/* kind: 0 sdh (regular), 1 dep-quant, 2 transform skip.  BITDEPTH 10: 15 - 10 = 5. */
static struct dq
derive_dq(int kind, int qp, int log2_w, int log2_h)
{
    struct dq d;
    int l2s = log2_w + log2_h;
    int shift, scale;
    if (kind == 2) {
        shift = 6 - qp / 6;
        scale = iq_scale[0][qp % 6];
    } else if (kind == 1) {
        shift = 6 + 1 - 5 - (qp + 1) / 6 + (l2s >> 1) + (l2s & 1);
        scale = iq_scale[l2s & 1][(qp + 1) % 6];
    } else {
        shift = 6 - 5 - qp / 6 + (l2s >> 1) + (l2s & 1);
        scale = iq_scale[l2s & 1][qp % 6];
    }
    d.scale = (int16_t)scale;
    d.neg   = shift < 0;
    d.shift = (uint8_t)(shift < 0 ? -shift : shift);
    return d;
}

/* ---------------------------------------------------------------- LFNST kernel choice */
static int
lfnst_set_of_mode(int m)   /* the 95-entry lfnst_mode_map as ranges */
{
    if (m <= 1)  return 0;
    if (m <= 12) return 1;
    if (m <= 23) return 2;
    if (m <= 44) return 3;
    if (m <= 55) return 2;
    return 1;
}

static int
wide_angle_mode(int log2_w, int log2_h, int mode)
{
    static const uint8_t mode_shift[6] = { 0, 6, 10, 12, 14, 15 };
    int d = log2_w - log2_h;
    int ms = mode_shift[d < 0 ? -d : d];
    if (log2_w > log2_h && mode < 2 + ms)        mode += 65;   /* VDIA - 1 */
    else if (log2_h > log2_w && mode > 66 - ms)  mode -= 67;   /* VDIA + 1 */
    return mode;
}

static int
lfnst_mode_luma(int log2_w, int log2_h, int intra_mode)
{
    if (intra_mode > 1) intra_mode = wide_angle_mode(log2_w, log2_h, intra_mode);
    if (intra_mode < 0)        intra_mode += 14 + 67;
    else if (intra_mode >= 67) intra_mode += 14;
    return intra_mode;
}

static uint8_t
lfnst_field(int mode_idx, int lfnst_idx)
{
    int transpose = (mode_idx < 67 && mode_idx > 34) || mode_idx >= 67 + 14;
    return (uint8_t)(1 | (lfnst_set_of_mode(mode_idx) << 1) | ((lfnst_idx & 1) << 3) | (transpose << 4));
}

/* ---------------------------------------------------------------- coefficient capture */
static uint64_t
valid_sb_mask(int log2_w, int log2_h)
{
    int nx = 1 << ((log2_w > 5 ? 5 : log2_w) - 2);
    int ny = 1 << ((log2_h > 5 ? 5 : log2_h) - 2);
    uint64_t row = (1ull << nx) - 1, m = 0;
    for (int i = 0; i < ny; ++i) m |= row << (i * 8);
    return m;
}

/* Copy the flagged 4x4 sub-blocks (reference layout: SB-major, SB (sx,sy) at
 * src[sy*4*stride + sx*16], stride = min(32, tb_w); rcn_dequant.c:160-236) into the arena,
 * 16 int16 per set bit in ascending bit order. */
static int
capture_sbs(ovhip_recorder *r, const int16_t *src, int log2_w, uint64_t map, uint32_t *off)
{
    int stride = 1 << (log2_w > 5 ? 5 : log2_w);
    int n = __builtin_popcountll(map);
    if (grow((void **)&r->coef, &r->cap_coef, r->n_coef + (size_t)n * 16, sizeof(int16_t))) return -1;
    *off = (uint32_t)r->n_coef;
    while (map) {
        int b = __builtin_ctzll(map);
        map &= map - 1;
        memcpy(r->coef + r->n_coef, src + (b >> 3) * 4 * stride + (b & 7) * 16, 16 * sizeof(int16_t));
        r->n_coef += 16;
    }
    return 0;
}

static int
capture_raster(ovhip_recorder *r, const int16_t *src, int n, uint32_t *off)
{
    size_t pad = (size_t)(n + 15) & ~(size_t)15;   /* keep every TB 32-byte aligned in the arena */
    if (grow((void **)&r->coef, &r->cap_coef, r->n_coef + pad, sizeof(int16_t))) return -1;
    *off = (uint32_t)r->n_coef;
    memcpy(r->coef + r->n_coef, src, (size_t)n * sizeof(int16_t));
    memset(r->coef + r->n_coef + n, 0, (pad - (size_t)n) * sizeof(int16_t));
    r->n_coef += pad;
    return 0;
}

static ovhip_tb_cmd *
new_tb(ovhip_recorder *r)
{
    if (grow((void **)&r->tb, &r->cap_tb, r->n_tb + 1, sizeof(ovhip_tb_cmd))) return NULL;
    ovhip_tb_cmd *c = &r->tb[r->n_tb++];
    memset(c, 0, sizeof(*c));
    c->plane2 = 0xff;
    c->c_scale = 1 << 11;
    c->tr_h = c->tr_v = OVHIP_DCT_II;
    return c;
}

/* ---------------------------------------------------------------- residual-add variants */
/* ict.ict[log2w][k] for k = 0,1,2 under rcn_init_ict_functions(type) (rcn_residuals.c:231-331) */
static uint8_t
ict_mode(int ict_type, int k)
{
    static const uint8_t tab[4][3] = {
        /* 0 */ { OVHIP_RES_ADD, OVHIP_RES_ADD, OVHIP_RES_ADD_HALF },
        /* 1 */ { OVHIP_RES_ADD | OVHIP_RES_SCALE, OVHIP_RES_ADD | OVHIP_RES_SCALE, OVHIP_RES_ADD_HALF | OVHIP_RES_SCALE },
        /* 2 */ { OVHIP_RES_ADD, OVHIP_RES_SUB, OVHIP_RES_SUB_HALF },
        /* 3 */ { OVHIP_RES_ADD | OVHIP_RES_SCALE, OVHIP_RES_SUB | OVHIP_RES_SCALE, OVHIP_RES_SUB_HALF | OVHIP_RES_SCALE },
    };
    return tab[ict_type & 3][k];
}

/* ---------------------------------------------------------------- one transform block */
struct tb_args {
    int plane, x, y, log2_w, log2_h;
    int is_luma;
    int tr_skip;            /* transform skip selected for this TB                    */
    int bdpcm;              /* 0 none, 1 horizontal (along a row), 2 vertical          */
    int qp, qp_skip;
    int lfnst_flag, lfnst_idx, lfnst_mode_idx;
    int cu_mts_flag, cu_mts_idx;
    int implicit_mts_ok;    /* luma: !is_mip && mts_implicit                           */
    uint16_t last_pos;
    uint64_t sig_sb_map;
    const int16_t *coef;
};

static int
emit_tb(ovhip_recorder *r, const ovhip_tu_state *st, const struct tb_args *a, ovhip_tb_cmd **out)
{
    ovhip_tb_cmd *c = new_tb(r);
    if (!c) return OVHIP_ENOMEM;
    c->x = (uint16_t)a->x; c->y = (uint16_t)a->y;
    c->plane = (uint8_t)a->plane;
    c->log2_w = (uint8_t)a->log2_w; c->log2_h = (uint8_t)a->log2_h;
    *out = c;

    int small = a->log2_w < 2 || a->log2_h < 2;        /* 2xN / Nx2 chroma TBs: raster storage */

    if (a->tr_skip && a->bdpcm) {
        /* rcn_bdpcm_tb (rcn_transform_tree.c:665-688): levels de-scanned only when the slice uses regular residual
         * coding for TS blocks AND the block has 4x4 sub-blocks; accumulated; then dequant_sb() per 16 samples */
        struct dq d = derive_dq(2, a->qp_skip, a->log2_w, a->log2_h);
        c->dq_scale = d.scale; c->dq_shift = d.shift; c->dq_neg = d.neg;
        c->tr_h = (uint8_t)(a->bdpcm - 1);
        const int n16 = (a->log2_w + a->log2_h) >= 4;
        if (st->sh_ts_disabled && !small) {
            c->kind = OVHIP_TB_TS | OVHIP_TB_FLAG_BDPCM;
            /* reorder_tb_4x4 takes the map as it is: a DC-only block with an empty map contributes nothing
             * (the "force the first sub-block" line is commented out there, rcn_transform_tree.c:143-144) */
            c->sig_sb_map = a->sig_sb_map & valid_sb_mask(a->log2_w, a->log2_h);
            return capture_sbs(r, a->coef, a->log2_w, c->sig_sb_map, &c->coef_off) ? OVHIP_ENOMEM : 0;
        }
        c->kind = (n16 ? OVHIP_TB_TS : OVHIP_TB_TS_RAW) | OVHIP_TB_FLAG_RASTER | OVHIP_TB_FLAG_BDPCM;
        return capture_raster(r, a->coef, 1 << (a->log2_w + a->log2_h), &c->coef_off) ? OVHIP_ENOMEM : 0;
    }
    if (a->tr_skip) {
        if (!st->sh_ts_disabled) {
            /* TS residual coding: coefficients are raster and final (memcpy in the reference) */
            c->kind = OVHIP_TB_TS_RAW | OVHIP_TB_FLAG_RASTER;
            return capture_raster(r, a->coef, 1 << (a->log2_w + a->log2_h), &c->coef_off) ? OVHIP_ENOMEM : 0;
        }
        struct dq d = derive_dq(2, a->qp_skip, a->log2_w, a->log2_h);
        c->dq_scale = d.scale; c->dq_shift = d.shift; c->dq_neg = d.neg;
        if (small) {
            /* memcpy + dequant_sb() per 16 coefficients (rcn_transform_tree.c:700-706): blocks
             * smaller than 16 samples (2x2, 2x4, 4x2) are therefore NOT de-quantised */
            c->kind = ((a->log2_w + a->log2_h) >= 4 ? OVHIP_TB_TS : OVHIP_TB_TS_RAW) | OVHIP_TB_FLAG_RASTER;
            return capture_raster(r, a->coef, 1 << (a->log2_w + a->log2_h), &c->coef_off) ? OVHIP_ENOMEM : 0;
        }
        c->kind = OVHIP_TB_TS;
        c->sig_sb_map = (a->sig_sb_map | !a->sig_sb_map) & valid_sb_mask(a->log2_w, a->log2_h);
        return capture_sbs(r, a->coef, a->log2_w, c->sig_sb_map, &c->coef_off) ? OVHIP_ENOMEM : 0;
    }

    struct dq d = derive_dq(st->dep_quant ? 1 : 0, a->qp, a->log2_w, a->log2_h);
    c->dq_scale = d.scale; c->dq_shift = d.shift; c->dq_neg = d.neg;

    if (small) {
        c->kind = OVHIP_TB_FLAG_RASTER;
        c->sig_sb_map = a->sig_sb_map;
        if (capture_raster(r, a->coef, 1 << (a->log2_w + a->log2_h), &c->coef_off)) return OVHIP_ENOMEM;
    } else {
        c->sig_sb_map = (a->sig_sb_map | !a->sig_sb_map) & valid_sb_mask(a->log2_w, a->log2_h);
        if (capture_sbs(r, a->coef, a->log2_w, c->sig_sb_map, &c->coef_off)) return OVHIP_ENOMEM;
    }

    int is_dc = !a->last_pos;
    if (a->is_luma) {
        if (a->implicit_mts_ok && !a->cu_mts_flag && (a->log2_w <= 4 || a->log2_h <= 4) && !a->lfnst_flag) {
            c->tr_h = a->log2_w <= 4 ? OVHIP_DST_VII : OVHIP_DCT_II;
            c->tr_v = a->log2_h <= 4 ? OVHIP_DST_VII : OVHIP_DCT_II;
            c->kind |= OVHIP_TB_TR;
        } else if (!a->cu_mts_flag) {
            if (a->lfnst_flag) {
                c->lfnst = lfnst_field(a->lfnst_mode_idx, a->lfnst_idx);
                is_dc = 0;
            }
            c->kind |= is_dc ? OVHIP_TB_DC : OVHIP_TB_TR;
        } else {
            c->tr_h = (uint8_t)(a->cu_mts_idx & 1);
            c->tr_v = (uint8_t)(a->cu_mts_idx >> 1);
            c->kind |= OVHIP_TB_TR;
        }
    } else {
        if (is_dc && !a->lfnst_flag) {
            c->kind |= OVHIP_TB_DC;
        } else {
            if (a->lfnst_flag) c->lfnst = lfnst_field(a->lfnst_mode_idx, a->lfnst_idx);
            c->kind |= OVHIP_TB_TR;
        }
    }
    return 0;
}

int
ovhip_rec_tu(ovhip_recorder *r, const ovhip_tu_state *st, const ovhip_tu_desc *tu)
{
    return ovhip_rec_tu_intra(r, st, tu, NULL, NULL);
}

/* ---- intra sub-partitions (recon_isp_subtree_v / _h, rcn_transform_tree.c:1087-1205) ---- */
void
ovhip_isp_geometry(int32_t log2_cb_w, int32_t log2_cb_h, int32_t vertical, int32_t *log2_pb, int32_t *n_pb, int32_t *log2_pred, int32_t *n_pred)
{
    const int l2s = vertical ? log2_cb_w : log2_cb_h, l2o = vertical ? log2_cb_h : log2_cb_w;    /* split / other side */
    int l2p = l2s - 2;
    if (l2o < 4 && l2p <= 4 - l2o) l2p = 4 - l2o;                 /* :1102-1104, :1172-1174: at least 16 samples per partition */
    *log2_pb = l2p; *n_pb = (1 << l2s) >> l2p;
    /* vertical partitions are predicted at least 4 columns at a time (:1127-1134); horizontal ones one by one */
    *log2_pred = (vertical && l2p < 2) ? 2 : l2p;
    *n_pred = (1 << l2s) >> *log2_pred;
}

int
ovhip_rec_isp_cu(ovhip_recorder *r, const ovhip_tu_state *st, const ovhip_isp_desc *cu)
{
    if (!r || !st || !cu || cu->log2_cb_w < 2 || cu->log2_cb_w > 6 || cu->log2_cb_h < 2 || cu->log2_cb_h > 6 || cu->log2_cb_w + cu->log2_cb_h < 5
        || (cu->cbf_mask && !cu->coef))
        return OVHIP_EINVAL;
    if (r->log) ovhip_calllog_isp_(r->log, st, cu);
    const size_t n0 = r->n_tb, c0 = r->n_coef, t0 = r->n_itask;
    int32_t l2p, n_pb, l2pred, n_pred;
    ovhip_isp_geometry(cu->log2_cb_w, cu->log2_cb_h, cu->vertical, &l2p, &n_pb, &l2pred, &n_pred);
    const int l2tw = cu->vertical ? l2p : cu->log2_cb_w, l2th = cu->vertical ? cu->log2_cb_h : l2p;     /* transform block */
    /* 64x2 partitions (a 64x8 CU split horizontally): the reference de-quantises them with a row stride of 32 and transforms with 64
     * (rcn_Xx2_tb, rcn_transform_tree.c:985-1009; dequant_tb :119-132) -- its result depends on memory nothing wrote, so there is no
     * reference result to be equal to.  They are reconstructed as H.266 defines them (8.7.4: a 64-point transform has 32 coded
     * columns, the rest is zero): the two coded rows of 32 (what the parser delivers) are laid out as raster rows of 64.
     * PARITY UNPINNED for these blocks: checked against a restatement of the specification, tests/test_gpu_edge_cases.py */
    const int thin64 = l2tw == 6 && l2th == 1;
    const int pb = 1 << l2p;
    /* transform types (:1110-1111, :1180-1181) */
    const int long_dst = cu->mts_enabled;
    int type_h, type_v;
    if (cu->vertical) { type_h = long_dst && l2p <= 4 && l2p > 1; type_v = long_dst && cu->log2_cb_h <= 4; }
    else              { type_h = long_dst && cu->log2_cb_w <= 4;  type_v = long_dst && l2p <= 4 && l2p > 1; }
    const int lfnst_mode = lfnst_mode_luma(cu->log2_cb_w, cu->log2_cb_h, cu->intra_mode);       /* drv_lfnst_mode_l on the CU's shape (:1117) */
    int ret, ti = -1, n_added = 0;
    for (int i = 0; i < n_pb; ++i) {
        const int off = i * pb;
        const int x = cu->x0 + (cu->vertical ? off : 0), y = cu->y0 + (cu->vertical ? 0 : off);
        const int cbf = (cu->cbf_mask >> (n_pb - i - 1)) & 1;
        if (!cu->vertical || !(off & 3)) {
            /* prediction call: the task of this partition (or of this group of 4 columns) */
            const int k = cu->vertical ? off >> l2pred : i;
            ovhip_itask t;
            memset(&t, 0, sizeof(t));
            t.kind = OVHIP_IT_LUMA; t.x = (uint16_t)x; t.y = (uint16_t)y;
            t.log2_w = (uint8_t)(cu->vertical ? l2pred : cu->log2_cb_w); t.log2_h = (uint8_t)(cu->vertical ? cu->log2_cb_h : l2p);
            t.mode = cu->intra_mode;
            t.flags = (uint16_t)(OVHIP_IF_ISP | ((cu->corner[k] & 1) ? OVHIP_IF_CORNER : 0) | ((cu->corner[k] & 2) ? OVHIP_IF_CORNER_L : 0));
            t.avl_abv = cu->avl_abv[k]; t.avl_lft = cu->avl_lft[k];
            t.isp_log2_cb_w = cu->log2_cb_w; t.isp_log2_cb_h = cu->log2_cb_h;
            t.isp_off_x = (uint8_t)(cu->vertical ? off : 0); t.isp_off_y = (uint8_t)(cu->vertical ? 0 : off);
            t.isp_log2_pb = (uint8_t)(cu->vertical ? l2p : cu->log2_cb_w);
            /* at least one level after the CU's previous prediction call (it reads what that one and its residuals left) */
            const uint16_t prev_level = ti >= 0 ? r->itask[ti].level : 0;
            if ((ti = ovhip_rec_itask_add_(r, &t, prev_level)) < 0) { ret = ti; goto fail; }
        }
        if (!cbf) continue;
        ovhip_tb_cmd *c = new_tb(r);
        if (!c) { ret = OVHIP_ENOMEM; goto fail; }
        c->x = (uint16_t)x; c->y = (uint16_t)y; c->plane = 0; c->log2_w = (uint8_t)l2tw; c->log2_h = (uint8_t)l2th;
        const struct dq d = derive_dq(st->dep_quant ? 1 : 0, st->qp_y, l2tw, l2th);
        c->dq_scale = d.scale; c->dq_shift = d.shift; c->dq_neg = d.neg;
        const int16_t *src = cu->coef + ((size_t)i << (l2tw + l2th));
        c->tr_h = (uint8_t)(type_h ? OVHIP_DST_VII : OVHIP_DCT_II); c->tr_v = (uint8_t)(type_v ? OVHIP_DST_VII : OVHIP_DCT_II);
        c->kind = OVHIP_TB_TR;
        if (l2tw < 2 || l2th < 2) {
            /* rcn_2xX_tb / rcn_1xX_tb / rcn_Xx2_tb / rcn_Xx1_tb (:919-1061): the whole block raster, de-quantised as a whole */
            c->kind |= OVHIP_TB_FLAG_RASTER;
            c->sig_sb_map = cu->sig_sb_map[i];
            if (thin64) {
                int16_t wide[128];
                memset(wide, 0, sizeof(wide));
                memcpy(wide, src, 32 * sizeof(int16_t)); memcpy(wide + 64, src + 32, 32 * sizeof(int16_t));
                if (capture_raster(r, wide, 128, &c->coef_off)) { ret = OVHIP_ENOMEM; goto fail; }
            } else
            if (capture_raster(r, src, 1 << (l2tw + l2th), &c->coef_off)) { ret = OVHIP_ENOMEM; goto fail; }
        } else {
            /* rcn_isp_tu (:869-917): sub-block storage, optional LFNST (which forces DCT-II), never the DC shortcut */
            c->sig_sb_map = (cu->sig_sb_map[i] | !cu->sig_sb_map[i]) & valid_sb_mask(l2tw, l2th);
            if (capture_sbs(r, src, l2tw, c->sig_sb_map, &c->coef_off)) { ret = OVHIP_ENOMEM; goto fail; }
            if (cu->lfnst_flag) { c->lfnst = lfnst_field(lfnst_mode, cu->lfnst_idx); c->tr_h = c->tr_v = OVHIP_DCT_II; }
        }
        c->res_mode = OVHIP_RES_ADD | OVHIP_RES_STORE;
        c->plane2 = 0xff;
        r->itask[ti].flags |= OVHIP_IF_RES_Y;
        r->itask[ti].isp_res_mask |= (uint8_t)(1u << (cu->vertical ? (off & 3) >> l2p : 0));
        ++n_added;
    }
    return n_added;
fail:
    r->n_tb = n0; r->n_coef = c0; r->n_itask = t0;
    return ret;
}

int
ovhip_rec_tu_intra(ovhip_recorder *r, const ovhip_tu_state *st, const ovhip_tu_desc *tu, const ovhip_itask *intra_l, const ovhip_itask *intra_c)
{
    if (!r || !st || !tu) return OVHIP_EINVAL;
    if (r->log) ovhip_calllog_tu_(r->log, st, tu, intra_l, intra_c);
    const size_t n0 = r->n_tb, c0 = r->n_coef, t0 = r->n_itask;
    int ret;
    ovhip_tb_cmd *c;
    int ti_l = -1, ti_c = -1;
    if ((intra_l && (intra_l->kind != OVHIP_IT_LUMA || tu->tree == 2)) || (intra_c && (intra_c->kind != OVHIP_IT_CHROMA || tu->tree == 1)))
        return OVHIP_EINVAL;
    /* the luma prediction comes first (rcn_intra_tu before rcn_tu_st, rcn_transform_tree.c:1437-1441) */
    if (intra_l) {
        if ((ti_l = ovhip_rec_itask_add_(r, intra_l, 0)) < 0) return ti_l;
    }

    /* ---- luma (rcn_tu_st / rcn_tu_l) ---- */
    if (tu->tree != 2 && (tu->cbf_mask & 0x10)) {
        struct tb_args a;
        memset(&a, 0, sizeof(a));
        int is_intra = !!(tu->cu_flags & CUF_PRED_MODE_INTRA);
        int is_mip = !!(tu->cu_flags & CUF_MIP) || !is_intra;
        a.plane = 0; a.x = tu->x0; a.y = tu->y0; a.log2_w = tu->log2_tb_w; a.log2_h = tu->log2_tb_h;
        a.is_luma = 1;
        a.tr_skip = !!(tu->tr_skip_mask & 0x10);
        a.bdpcm = (tu->cu_flags & CUF_BDPCM_LUMA) ? 1 + !!(tu->cu_flags & CUF_BDPCM_LUMA_DIR) : 0;   /* rcn_transform_skip_tb_l */
        a.qp = st->qp_y; a.qp_skip = st->qp_y_skip;
        a.lfnst_flag = tu->lfnst_flag; a.lfnst_idx = tu->lfnst_idx;
        a.lfnst_mode_idx = lfnst_mode_luma(tu->log2_tb_w, tu->log2_tb_h, is_mip ? 0 : st->intra_mode);
        a.cu_mts_flag = tu->cu_mts_flag; a.cu_mts_idx = tu->cu_mts_idx;
        a.implicit_mts_ok = !is_mip && st->mts_implicit;
        a.last_pos = tu->last_pos[2]; a.sig_sb_map = tu->sig_sb_map[2]; a.coef = tu->coef[2];
        if ((ret = emit_tb(r, st, &a, &c))) goto fail;
        c->res_mode = OVHIP_RES_ADD;
        if (ti_l >= 0) { c->res_mode |= OVHIP_RES_STORE; r->itask[ti_l].flags |= OVHIP_IF_RES_Y; }
    }

    /* ---- chroma ---- */
    if (tu->tree != 1 && ((tu->cbf_mask & 0xb) || intra_c)) {
        int xc, yc, l2w, l2h, lfnst_flag;
        if (tu->tree == 2) { xc = tu->x0; yc = tu->y0; l2w = tu->log2_tb_w; l2h = tu->log2_tb_h; lfnst_flag = tu->lfnst_flag; }
        else { xc = tu->x0 >> 1; yc = tu->y0 >> 1; l2w = tu->log2_tb_w - 1; l2h = tu->log2_tb_h - 1; lfnst_flag = 0; }
        int cbf_c = tu->cbf_mask & 0x3;
        int16_t scale = st->lmcs_scale_c ? st->lmcs_chroma_scale : (int16_t)(1 << 11);
        /* lmcs_scale_c == 2: the scale is derived on the device; the command carries the region index */
        const int indirect = st->lmcs_scale_c == 2;
        uint16_t reg_lvl = 0;
        if (indirect) {
            if (!r->n_reg) { ret = OVHIP_EINVAL; goto fail; }
            scale = (int16_t)(r->n_reg - 1);
            reg_lvl = r->reg_level ? r->reg_level[r->n_reg - 1] : 0;
        }
        /* the chroma prediction of an intra CU sits between the TU's luma and chroma residuals (rcn_tu_st :1269-1287);
         * a chroma block of any other CU becomes ordered when its residual scale comes from an ordered region */
        {
            const int has_res = (tu->cbf_mask & 0xb) != 0;
            const int scaled = (l2w + l2h != 2) && (st->ict_type & 1) && has_res;
            if (intra_c) {
                if ((ti_c = ovhip_rec_itask_add_(r, intra_c, scaled && indirect ? (uint16_t)(reg_lvl ? reg_lvl : 0) : 0)) < 0) { ret = ti_c; goto fail; }
            } else if (scaled && indirect && reg_lvl && has_res) {
                ovhip_itask t;
                memset(&t, 0, sizeof(t));
                t.x = (uint16_t)xc; t.y = (uint16_t)yc; t.log2_w = (uint8_t)l2w; t.log2_h = (uint8_t)l2h; t.kind = OVHIP_IT_RES_C;
                if ((ti_c = ovhip_rec_itask_add_(r, &t, reg_lvl)) < 0) { ret = ti_c; goto fail; }
            }
        }
        struct tb_args a;
        memset(&a, 0, sizeof(a));
        a.x = xc; a.y = yc; a.log2_w = l2w; a.log2_h = l2h;
        a.lfnst_flag = lfnst_flag; a.lfnst_idx = tu->lfnst_idx; a.lfnst_mode_idx = st->lfnst_mode_c;
        a.bdpcm = (tu->cu_flags & CUF_BDPCM_CHROMA) ? 1 + !!(tu->cu_flags & CUF_BDPCM_CHROMA_DIR) : 0;   /* rcn_transform_skip_tb_c */

        if (tu->cbf_mask & 0x8) {
            /* joint CbCr: ONE residual (coded in the Cb slot), applied to both planes */
            int first, second, k2;
            a.tr_skip = !!(tu->tr_skip_mask & 0x1);
            a.qp      = cbf_c == 3 ? st->qp_jcbcr : cbf_c == 1 ? st->qp_cr : st->qp_cb;
            a.qp_skip = cbf_c == 3 ? st->qp_jcbcr_skip : cbf_c == 2 ? st->qp_cb_skip : st->qp_cr_skip;
            a.last_pos = tu->last_pos[0]; a.sig_sb_map = tu->sig_sb_map[0]; a.coef = tu->coef[0];
            if (cbf_c == 3)      { first = 1; second = 2; k2 = 1; }
            else if (cbf_c == 2) { first = 1; second = 2; k2 = 2; }
            else                 { first = 2; second = 1; k2 = 2; }
            a.plane = first;
            if ((ret = emit_tb(r, st, &a, &c))) goto fail;
            c->c_scale   = (l2w + l2h == 2) ? (int16_t)(1 << 11) : scale;
            c->res_mode  = ict_mode(st->ict_type, 0);
            c->plane2    = (uint8_t)second;
            c->res_mode2 = ict_mode(st->ict_type, k2);
            if (indirect && l2w + l2h != 2 && (c->res_mode & OVHIP_RES_SCALE)) { c->res_mode |= OVHIP_RES_SCALE_IDX; c->res_mode2 |= OVHIP_RES_SCALE_IDX; }
            if (ti_c >= 0) {
                ovhip_itask *t = &r->itask[ti_c];
                t->flags |= OVHIP_IF_RES_CB | OVHIP_IF_RES_CR;
                if (c->res_mode & OVHIP_RES_SCALE) { t->flags |= OVHIP_IF_RES_SCALE | (indirect ? OVHIP_IF_SCALE_IDX : 0); t->c_scale = c->c_scale; }
                c->res_mode |= OVHIP_RES_STORE; c->res_mode2 |= OVHIP_RES_STORE;
            }
        } else {
            for (int comp = 0; comp < 2; ++comp) {      /* 0: Cb (cbf 0x2), 1: Cr (cbf 0x1) */
                int bit = comp ? 0x1 : 0x2;
                if (!(cbf_c & bit)) continue;
                a.plane = 1 + comp;
                a.tr_skip = !!(tu->tr_skip_mask & bit);
                a.qp      = comp ? st->qp_cr : st->qp_cb;
                a.qp_skip = comp ? st->qp_cr_skip : st->qp_cb_skip;
                a.last_pos = tu->last_pos[comp]; a.sig_sb_map = tu->sig_sb_map[comp]; a.coef = tu->coef[comp];
                if ((ret = emit_tb(r, st, &a, &c))) goto fail;
                if (l2w + l2h > 2) {
                    c->res_mode = ict_mode(st->ict_type, 0); c->c_scale = scale;
                    if (indirect && (c->res_mode & OVHIP_RES_SCALE)) c->res_mode |= OVHIP_RES_SCALE_IDX;
                }
                else               { c->res_mode = OVHIP_RES_ADD; }
                if (ti_c >= 0) {
                    ovhip_itask *t = &r->itask[ti_c];
                    t->flags |= comp ? OVHIP_IF_RES_CR : OVHIP_IF_RES_CB;
                    if (c->res_mode & OVHIP_RES_SCALE) { t->flags |= OVHIP_IF_RES_SCALE | (indirect ? OVHIP_IF_SCALE_IDX : 0); t->c_scale = c->c_scale; }
                    c->res_mode |= OVHIP_RES_STORE;
                }
            }
        }
    }
    return (int)(r->n_tb - n0);
fail:
    r->n_tb = n0;
    r->n_coef = c0;
    r->n_itask = t0;          /* (the level maps keep the marks of the dropped tasks: harmless, levels only grow) */
    return ret;
}

/* ---------------------------------------------------------------- transform tree of one CU
 * rcn_transform_tree + rcn_res_wrap (rcn_transform_tree.c:1432-1506). */
static int
tt_walk(ovhip_recorder *r, const ovhip_tu_state *st, const ovhip_tt_desc *tt, int x0, int y0, int log2_w, int log2_h,
        int depth, const ovhip_tu_info *ti)
{
    const int max = tt->log2_max_tb_s;
    const int split_v = log2_w > max, split_h = log2_h > max;
    const int nsub = depth ? 1 : (1 << (split_v + split_h));
    int n = 0, k;

    /* a 128-sample side next to a shorter one is first cut into two 64-sample halves: TUInfo slots 0 and 8 */
    if (log2_w > 6 && log2_h < 7) {
        if ((k = tt_walk(r, st, tt, x0, y0, 6, log2_h, depth + 1, ti)) < 0) return k;
        n += k;
        if ((k = tt_walk(r, st, tt, x0 + 64, y0, 6, log2_h, depth + 1, ti + 8)) < 0) return k;
        return n + k;
    }
    if (log2_h > 6 && log2_w < 7) {
        if ((k = tt_walk(r, st, tt, x0, y0, log2_w, 6, depth + 1, ti)) < 0) return k;
        n += k;
        if ((k = tt_walk(r, st, tt, x0, y0 + 64, log2_w, 6, depth + 1, ti + 8)) < 0) return k;
        return n + k;
    }
    if (split_v || split_h) {
        const int w1 = (1 << log2_w) >> split_v, h1 = (1 << log2_h) >> split_h;
        const int l2w1 = log2_w - split_v, l2h1 = log2_h - split_h;
        if ((k = tt_walk(r, st, tt, x0, y0, l2w1, l2h1, depth + 1, ti)) < 0) return k;
        n += k;
        if (split_v) { if ((k = tt_walk(r, st, tt, x0 + w1, y0, l2w1, l2h1, depth + 1, ti + 1 * nsub)) < 0) return k; n += k; }
        if (split_h) { if ((k = tt_walk(r, st, tt, x0, y0 + h1, l2w1, l2h1, depth + 1, ti + 2 * nsub)) < 0) return k; n += k; }
        if (split_h && split_v) { if ((k = tt_walk(r, st, tt, x0 + w1, y0 + h1, l2w1, l2h1, depth + 1, ti + 3 * nsub)) < 0) return k; n += k; }
        return n;
    }
    /* leaf: rcn_res_wrap -> rcn_tu_st / rcn_tu_l / rcn_tu_c */
    ovhip_tu_desc d;
    memset(&d, 0, sizeof(d));
    d.x0 = (uint16_t)x0; d.y0 = (uint16_t)y0; d.log2_tb_w = (uint8_t)log2_w; d.log2_tb_h = (uint8_t)log2_h;
    d.tree = tt->tree; d.cbf_mask = ti->cbf_mask; d.cu_flags = tt->cu_flags;
    d.tr_skip_mask = ti->tr_skip_mask; d.cu_mts_flag = ti->cu_mts_flag; d.cu_mts_idx = ti->cu_mts_idx;
    d.lfnst_flag = ti->lfnst_flag; d.lfnst_idx = ti->lfnst_idx;
    for (int c = 0; c < 3; ++c) {
        d.last_pos[c] = ti->last_pos[c]; d.sig_sb_map[c] = ti->sig_sb_map[c];
        d.coef[c] = tt->residual[c] ? tt->residual[c] + ti->pos_offset : NULL;
    }
    if (!d.cbf_mask) return 0;
    return ovhip_rec_tu(r, st, &d);
}

int
ovhip_rec_transform_tree(ovhip_recorder *r, const ovhip_tu_state *st, const ovhip_tt_desc *tt)
{
    if (!r || !st || !tt || !tt->tu_info || tt->tree > 2 || tt->log2_w > 7 || tt->log2_h > 7 || tt->log2_max_tb_s > 6) return OVHIP_EINVAL;
    const size_t n0 = r->n_tb, c0 = r->n_coef;
    const int n = tt_walk(r, st, tt, tt->x0, tt->y0, tt->log2_w, tt->log2_h, 0, tt->tu_info);
    if (n < 0) { r->n_tb = n0; r->n_coef = c0; }
    return n;
}

/* ---------------------------------------------------------------- prediction units */
static int32_t clip3(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : v > hi ? hi : v; }

static void
clip_mv(const ovhip_recorder *r, int px, int py, int pw, int ph, int32_t *mvx, int32_t *mvy)
{
    /* clip_mv(): keeps the reference window within [-(pb+3), pic+2] of the block position */
    *mvx = clip3(*mvx, -((pw + 3 + px) << 4), (r->pic_w + 2 - px) << 4);
    *mvy = clip3(*mvy, -((ph + 3 + py) << 4), (r->pic_h + 2 - py) << 4);
}

/* bdof_enable / dmvr_enable CUs: the reference's caller cuts the CU into <=16x16 blocks and hands each
 * to rcn_bdof_mcp_l (+ one rcn_mcp_b_c for the CU's chroma) or rcn_dmvr_mv_refine
 * (vcl_coding_unit.c:2450-2472, :2598-2668). */
static int
rec_pu_refined(ovhip_recorder *r, const ovhip_pu_desc *pu)
{
    const int pw = 1 << pu->log2_w, ph = 1 << pu->log2_h;
    const int uw = pw > 16 ? 16 : pw, uh = ph > 16 ? 16 : ph;
    const int dmvr = (pu->refine & OVHIP_PU_DMVR) != 0;
    int n = 0;
    if ((pu->inter_dir & 3) != 3 || pw < 8 || ph < 8 || pw * ph < 128) return OVHIP_EINVAL;   /* check_bdof() */

    uint8_t flags = (pu->refine & OVHIP_PU_BDOF) ? OVHIP_MC_BDOF : 0;
    if (dmvr) flags |= OVHIP_MC_DMVR;
    if (pu->prec_amvr_half) flags |= OVHIP_MC_HPEL_FILT;
    if (pu->lmcs)           flags |= OVHIP_MC_LMCS;

    /* chroma of a BDOF-only CU: rcn_mcp_b_c clips the MVs against the CU, not the 16x16 block */
    int32_t c0x = pu->mv0x, c0y = pu->mv0y, c1x = pu->mv1x, c1y = pu->mv1y;
    clip_mv(r, pu->x0, pu->y0, pw, ph, &c0x, &c0y);
    clip_mv(r, pu->x0, pu->y0, pw, ph, &c1x, &c1y);

    /* ... and still applies its identical-motion shortcut (rcn_inter.c:2935-2951; never true for a
     * conforming bdof_enable, whose references lie on opposite sides of the current picture) */
    const int ident = pu->poc0 == pu->poc1 && pu->mv0x == pu->mv1x && pu->mv0y == pu->mv1y;

    for (int uy = 0; uy < ph; uy += uh) {
        for (int ux = 0; ux < pw; ux += uw) {
            ovhip_mc_unit u;
            memset(&u, 0, sizeof(u));
            u.x = (uint16_t)(pu->x0 + ux); u.y = (uint16_t)(pu->y0 + uy);
            u.w = (uint8_t)uw; u.h = (uint8_t)uh;
            u.dir = 3; u.flags = flags;
            u.ref0 = pu->ref0; u.ref1 = pu->ref1;
            u.w0 = u.w1 = 4;
            u.mv0x = pu->mv0x; u.mv0y = pu->mv0y; u.mv1x = pu->mv1x; u.mv1y = pu->mv1y;
            int split_chroma = 0;
            if (!dmvr) {
                clip_mv(r, u.x, u.y, uw, uh, &u.mv0x, &u.mv0y);       /* rcn_bdof_mcp_l, rcn_inter.c:1166-1170 */
                clip_mv(r, u.x, u.y, uw, uh, &u.mv1x, &u.mv1y);
                split_chroma = ident || u.mv0x != c0x || u.mv0y != c0y || u.mv1x != c1x || u.mv1y != c1y;
                if (split_chroma) u.flags |= OVHIP_MC_NO_CHROMA;
                if (!(pu->planes & 2)) { u.flags |= OVHIP_MC_NO_CHROMA; split_chroma = 0; }   /* rcn_bdof_mcp_l alone */
            }
            if (grow((void **)&r->mcx, &r->cap_mcx, r->n_mcx + 1, sizeof(u))) return OVHIP_ENOMEM;
            r->mcx[r->n_mcx++] = u;
            ++n;
            if (split_chroma) {
                /* the two clips disagree (block far outside the picture): chroma as a plain unit */
                u.flags = (uint8_t)((flags & OVHIP_MC_HPEL_FILT) | OVHIP_MC_NO_LUMA);
                if (ident) u.dir = 2;
                u.mv0x = c0x; u.mv0y = c0y; u.mv1x = c1x; u.mv1y = c1y;
                if (grow((void **)&r->mc, &r->cap_mc, r->n_mc + 1, sizeof(u))) return OVHIP_ENOMEM;
                r->mc[r->n_mc++] = u;
                ++n;
            }
        }
    }
    return n;
}

/* rcn_gpm_b (rcn_inter.c:3118-3143) -> rcn_mc_rpr_b_l/_c with gpm_ctx: two uni-predictions of the whole CU
 * (rcn_mcp_bidir0_l/_c: clip_mv against the CU) blended by put_weighted_gpm_bi_pixels with the weight plane
 * of rcn_gpm_weights_and_steps.  The reference walks mirrored pre-stored masks (rcn_gpm.c:149-205); the plane
 * is affine in the sample position, so the command carries it in closed form (H.266 8.5.7.2):
 *   weightIdx = ((x + offX) * 2 + 1) * dis[angle] + ((y + offY) * 2 + 1) * dis[angle + 8]
 *   w = clip3(0, 8, ((partFlip ? 32 + weightIdx : 32 - weightIdx) + 4) >> 3) */
#include "vvc_gpm_tables.h"
static int
rec_pu_gpm(ovhip_recorder *r, const ovhip_pu_desc *pu)
{
    const int pw = 1 << pu->log2_w, ph = 1 << pu->log2_h;
    if (pu->gpm_split_dir > 63 || pu->log2_w < 3 || pu->log2_h < 3 || pu->log2_w > 6 || pu->log2_h > 6) return OVHIP_EINVAL;

    const int angle = ovt_gpm_params[pu->gpm_split_dir][0], dist = ovt_gpm_params[pu->gpm_split_dir][1];
    const int dx = ovt_gpm_dis[angle], dy = ovt_gpm_dis[(angle + 8) & 31];
    const int flip = (angle >= 13 && angle <= 27) ? 0 : 1;
    const int shift_hor = (angle % 16 == 8 || (angle % 16 != 0 && ph >= pw)) ? 0 : 1;
    int off_x = -(pw >> 1), off_y = -(ph >> 1);
    if (dist > 0) {
        if (!shift_hor) off_y += angle < 16 ? (dist * ph) >> 3 : -((dist * ph) >> 3);
        else            off_x += angle < 16 ? (dist * pw) >> 3 : -((dist * pw) >> 3);
    }
    const int sgn = flip ? 1 : -1;
    const int A = sgn * 2 * dx, B = sgn * 2 * dy;
    const int K = 36 + sgn * ((2 * off_x + 1) * dx + (2 * off_y + 1) * dy);

    int32_t mv0x = pu->mv0x, mv0y = pu->mv0y, mv1x = pu->mv1x, mv1y = pu->mv1y;
    clip_mv(r, pu->x0, pu->y0, pw, ph, &mv0x, &mv0y);
    clip_mv(r, pu->x0, pu->y0, pw, ph, &mv1x, &mv1y);

    uint8_t flags = OVHIP_MC_GPM;
    if (pu->prec_amvr_half) flags |= OVHIP_MC_HPEL_FILT;
    if (pu->lmcs)           flags |= OVHIP_MC_LMCS;
    const int uw = pw > 16 ? 16 : pw, uh = ph > 16 ? 16 : ph;
    int n = 0;
    for (int uy = 0; uy < ph; uy += uh) {
        for (int ux = 0; ux < pw; ux += uw) {
            if (grow((void **)&r->mc, &r->cap_mc, r->n_mc + 1, sizeof(ovhip_mc_unit))) return OVHIP_ENOMEM;
            ovhip_mc_unit *u = &r->mc[r->n_mc++];
            memset(u, 0, sizeof(*u));
            u->x = (uint16_t)(pu->x0 + ux); u->y = (uint16_t)(pu->y0 + uy);
            u->w = (uint8_t)uw; u->h = (uint8_t)uh;
            u->dir = 3; u->flags = flags;
            u->ref0 = pu->ref0; u->ref1 = pu->ref1;
            u->w0 = u->w1 = 4;
            u->mv0x = mv0x; u->mv0y = mv0y; u->mv1x = mv1x; u->mv1y = mv1y;
            const int k = K + A * ux + B * uy;
            u->aux = ((uint32_t)k & 0xffff) | ((uint32_t)(A & 0xff) << 16) | ((uint32_t)(B & 0xff) << 24);
            ++n;
        }
    }
    return n;
}

int
ovhip_rec_cu_inter(ovhip_recorder *r, const ovhip_pu_desc *pu, const ovhip_affine_desc *aff)
{
    if (!r || !pu == !aff) return OVHIP_EINVAL;
    return pu ? ovhip_rec_pu(r, pu) : ovhip_rec_affine_cu(r, aff);
}

int
ovhip_rec_pu(ovhip_recorder *r, const ovhip_pu_desc *pu)
{
    if (r->log) ovhip_calllog_pu_(r->log, pu);
    int pw = 1 << pu->log2_w, ph = 1 << pu->log2_h;
    int dir = pu->inter_dir & 3;
    if (!dir) return OVHIP_EINVAL;
    if (pu->ciip_wt > 3 || (pu->ciip_wt && pu->refine)) return OVHIP_EINVAL;
    if (pu->refine & OVHIP_PU_GPM) return rec_pu_gpm(r, pu);
    if (pu->refine) return rec_pu_refined(r, pu);

    /* rcn_mcp_b: bi with identical motion degenerates to uni-pred from list 1 */
    if (dir == 3 && pu->poc0 == pu->poc1 && pu->mv0x == pu->mv1x && pu->mv0y == pu->mv1y) dir = 2;
    else if (dir != 3 && (dir & 2)) dir = 2;

    /* clip_mv(): keeps the reference window within [-(pb+3), pic+2] of the PU position */
    int32_t x_max = (r->pic_w + 2 - pu->x0) << 4, y_max = (r->pic_h + 2 - pu->y0) << 4;
    int32_t x_min = -((pw + 3 + pu->x0) << 4),    y_min = -((ph + 3 + pu->y0) << 4);
    int32_t mv0x = clip3(pu->mv0x, x_min, x_max), mv0y = clip3(pu->mv0y, y_min, y_max);
    int32_t mv1x = clip3(pu->mv1x, x_min, x_max), mv1y = clip3(pu->mv1y, y_min, y_max);
    /* the list a uni-predicted unit does not use: the caller's fields hold whatever was there (found with AddressSanitizer's malloc
     * fill: the same parse recorded different bytes) -- nothing reads them, but what is uploaded is a function of the stream only */
    uint8_t ref0 = pu->ref0, ref1 = pu->ref1;
    if (!(dir & 1)) { mv0x = mv0y = 0; ref0 = 0; }
    if (!(dir & 2)) { mv1x = mv1y = 0; ref1 = 0; }

    int8_t w0 = 4, w1 = 4;
    if (dir == 3 && pu->bcw_idx_plus1 != 0 && pu->bcw_idx_plus1 != 3) {
        static const int8_t bcw[5] = { -2, 3, 4, 5, 10 };
        if (pu->bcw_idx_plus1 > 5) return OVHIP_EINVAL;
        w1 = bcw[pu->bcw_idx_plus1 - 1];
        w0 = (int8_t)(8 - w1);
    }

    uint8_t flags = 0;
    if (pu->prec_amvr_half) flags |= OVHIP_MC_HPEL_FILT;
    if (pw == 4 && ph == 4) flags |= OVHIP_MC_FILT_4x4;
    if (!(pu->planes & 1))  flags |= OVHIP_MC_NO_LUMA;
    if (!(pu->planes & 2))  flags |= OVHIP_MC_NO_CHROMA;
    if (pu->lmcs)           flags |= OVHIP_MC_LMCS;

    int uw = pw > 16 ? 16 : pw, uh = ph > 16 ? 16 : ph;
    int nu = (pw / uw) * (ph / uh);
    if (grow((void **)&r->mc, &r->cap_mc, r->n_mc + (size_t)nu, sizeof(ovhip_mc_unit))) return OVHIP_ENOMEM;
    for (int uy = 0; uy < ph; uy += uh) {
        for (int ux = 0; ux < pw; ux += uw) {
            ovhip_mc_unit *u = &r->mc[r->n_mc++];
            memset(u, 0, sizeof(*u));
            u->x = (uint16_t)(pu->x0 + ux); u->y = (uint16_t)(pu->y0 + uy);
            u->w = (uint8_t)uw; u->h = (uint8_t)uh;
            u->dir = (uint8_t)dir; u->flags = flags;
            u->ref0 = ref0; u->ref1 = ref1;
            u->w0 = w0; u->w1 = w1;
            u->mv0x = mv0x; u->mv0y = mv0y; u->mv1x = mv1x; u->mv1y = mv1y;
            /* fused CIIP blend (rcn_ciip_weighted_sum): chroma of a CU 4 luma samples wide keeps the inter prediction */
            if (pu->ciip_wt) u->aux = (uint32_t)(pu->ciip_wt & 7) | (pu->log2_w <= 2 ? 0x100u : 0u);
        }
    }
    return nu;
}

/* ---------------------------------------------------------------- affine CUs
 * rcn_affine_mcp_b_l / rcn_affine_prof_mcp_b_l / rcn_affine_mcp_b_c (drv_affine_mvp.c:3264-3411):
 * every 4x4 luma sub-block is predicted with its own motion vector, every 4x4 chroma block with
 * the average of the top-left and bottom-right sub-block vectors of its 8x8 luma area. */
int
ovhip_rec_affine_cu(ovhip_recorder *r, const ovhip_affine_desc *cu)
{
    const int cw = 1 << cu->log2_w, ch = 1 << cu->log2_h;
    int dir = cu->inter_dir & 3;
    if (!dir || cw < 8 || ch < 8 || !cu->mv0 || !cu->mv1 || cu->mv_stride < (cw >> 2)) return OVHIP_EINVAL;
    if (r->log) ovhip_calllog_affine_(r->log, cu);
    if (dir != 3 && (dir & 2)) dir = 2;

    int8_t w0 = 4, w1 = 4;
    if (dir == 3 && cu->bcw_idx_plus1 != 0 && cu->bcw_idx_plus1 != 3) {
        static const int8_t bcw[5] = { -2, 3, 4, 5, 10 };
        if (cu->bcw_idx_plus1 > 5) return OVHIP_EINVAL;
        w1 = bcw[cu->bcw_idx_plus1 - 1];
        w0 = (int8_t)(8 - w1);
    }

    uint32_t prof_off = 0;
    if (cu->prof_dir) {
        if (grow((void **)&r->aff_side, &r->cap_side, r->n_side + 32, sizeof(int32_t))) return OVHIP_ENOMEM;
        prof_off = (uint32_t)r->n_side;
        memcpy(r->aff_side + r->n_side, cu->dmv_scale, 128);
        /* h / v tables of a list PROF is not applied to: uninitialised in the caller (compute_prof_dmv_scale runs per refined
         * list, drv_affine_mvp.c:3325-3340); never read on the device, recorded as zeros */
        if (!(cu->prof_dir & 1)) memset(r->aff_side + r->n_side, 0, 64);
        if (!(cu->prof_dir & 2)) memset(r->aff_side + r->n_side + 16, 0, 64);
        r->n_side += 32;
    }

    const int uw = cw > 16 ? 16 : cw, uh = ch > 16 ? 16 : ch;
    int n = 0;
    for (int uy = 0; uy < ch; uy += uh) {
        for (int ux = 0; ux < cw; ux += uw) {
            const int nl = (uw >> 2) * (uh >> 2), nc = (uw >> 3) * (uh >> 3);
            if (grow((void **)&r->aff, &r->cap_aff, r->n_aff + 1, sizeof(ovhip_aff_unit))) return OVHIP_ENOMEM;
            if (grow((void **)&r->aff_side, &r->cap_side, r->n_side + 4 * (size_t)(nl + nc), sizeof(int32_t))) return OVHIP_ENOMEM;
            ovhip_aff_unit *u = &r->aff[r->n_aff++];
            memset(u, 0, sizeof(*u));
            u->x = (uint16_t)(cu->x0 + ux); u->y = (uint16_t)(cu->y0 + uy);
            u->w = (uint8_t)uw; u->h = (uint8_t)uh;
            u->dir = (uint8_t)dir;
            u->flags = (uint8_t)((cu->prof_dir ? OVHIP_AFF_PROF : 0) | (cu->lmcs ? OVHIP_AFF_LMCS : 0));
            u->ref0 = cu->ref0; u->ref1 = cu->ref1;
            u->w0 = w0; u->w1 = w1;
            u->prof_dir = cu->prof_dir;
            u->side_off = (uint32_t)r->n_side;
            u->prof_off = prof_off;
            int32_t *o = r->aff_side + r->n_side;
            for (int sy = 0; sy < uh; sy += 4) {
                for (int sx = 0; sx < uw; sx += 4) {
                    const int k = ((uy + sy) >> 2) * cu->mv_stride + ((ux + sx) >> 2);
                    int32_t m[4] = { cu->mv0[2 * k], cu->mv0[2 * k + 1], cu->mv1[2 * k], cu->mv1[2 * k + 1] };
                    /* the list a uni-predicted CU does not use: whatever the caller's OVMV held (found by the chained stream
                     * fixture: stack words of the reference's affine drivers) -- never read on the device, never recorded */
                    if (!(dir & 1)) m[0] = m[1] = 0;
                    if (!(dir & 2)) m[2] = m[3] = 0;
                    /* rcn_mcp_b_l's identical-motion shortcut; rcn_prof_mcp_b_l has none (rcn_inter.c:2864-2918) */
                    if (!cu->prof_dir && dir == 3 && cu->poc0 == cu->poc1 && m[0] == m[2] && m[1] == m[3])
                        u->ident_l |= (uint16_t)(1u << ((sy >> 2) * (uw >> 2) + (sx >> 2)));
                    clip_mv(r, u->x + sx, u->y + sy, 4, 4, &m[0], &m[1]);
                    clip_mv(r, u->x + sx, u->y + sy, 4, 4, &m[2], &m[3]);
                    memcpy(o, m, sizeof(m));
                    o += 4;
                }
            }
            for (int sy = 0; sy < uh; sy += 8) {
                for (int sx = 0; sx < uw; sx += 8) {
                    const int k = ((uy + sy) >> 2) * cu->mv_stride + ((ux + sx) >> 2), k2 = k + cu->mv_stride + 1;
                    int32_t m[4] = { 0, 0, 0, 0 };
                    if (dir & 1) { m[0] = cu->mv0[2 * k] + cu->mv0[2 * k2]; m[1] = cu->mv0[2 * k + 1] + cu->mv0[2 * k2 + 1]; }
                    if (dir & 2) { m[2] = cu->mv1[2 * k] + cu->mv1[2 * k2]; m[3] = cu->mv1[2 * k + 1] + cu->mv1[2 * k2 + 1]; }
                    for (int c = 0; c < 4; ++c) { m[c] += m[c] < 0; m[c] >>= 1; }
                    if (dir == 3 && cu->poc0 == cu->poc1 && m[0] == m[2] && m[1] == m[3])
                        u->ident_c |= (uint8_t)(1u << ((sy >> 3) * (uw >> 3) + (sx >> 3)));
                    clip_mv(r, u->x + sx, u->y + sy, 8, 8, &m[0], &m[1]);
                    clip_mv(r, u->x + sx, u->y + sy, 8, 8, &m[2], &m[3]);
                    memcpy(o, m, sizeof(m));
                    o += 4;
                }
            }
            r->n_side += 4 * (size_t)(nl + nc);
            ++n;
        }
    }
    return n;
}

/* ---------------------------------------------------------------- LMCS chroma-scale regions */
static int bit_length(uint32_t v) { int n = 0; while (v) { ++n; v >>= 1; } return n; }

int
ovhip_rec_lmcs_region(ovhip_recorder *r, int32_t x0, int32_t y0, uint32_t abv_mask, uint32_t lft_mask)
{
    if (x0 < 0 || y0 < 0 || x0 >= r->pic_w || y0 >= r->pic_h || r->n_reg >= 32767) return OVHIP_EINVAL;
    if (r->log) ovhip_calllog_region_(r->log, x0, y0, abv_mask, lft_mask);
    if (grow((void **)&r->reg, &r->cap_reg, r->n_reg + 1, sizeof(*r->reg))) return OVHIP_ENOMEM;
    ovhip_lmcs_region *g = &r->reg[r->n_reg];
    memset(g, 0, sizeof(*g));
    g->x = (uint16_t)x0; g->y = (uint16_t)y0;
    /* lmcs_compute_luma_average walks each mask until it is exhausted (rcn_lmcs.c:219-246) */
    g->n_abv = (uint8_t)bit_length(abv_mask & 0xffff);
    g->n_lft = (uint8_t)bit_length(lft_mask & 0xffff);
    /* luma around the region reconstructed by ordered tasks: the scale is derived in the ordered pass */
    if (r->cap_reglvl < r->cap_reg) {
        uint16_t *q = (uint16_t *)realloc(r->reg_level, r->cap_reg * sizeof(uint16_t));
        if (!q) return OVHIP_ENOMEM;
        r->reg_level = q; r->cap_reglvl = r->cap_reg;
    }
    const uint16_t lvl = ovhip_rec_region_level_(r, x0, y0, g->n_abv, g->n_lft);
    r->reg_level[r->n_reg] = lvl;
    if (lvl) {
        ovhip_itask t;
        memset(&t, 0, sizeof(t));
        t.x = g->x; t.y = g->y; t.kind = OVHIP_IT_REGION; t.c_scale = (int16_t)r->n_reg;
        int ti = ovhip_rec_itask_add_(r, &t, (uint16_t)(lvl - 1));
        if (ti < 0) return ti;
        g->ordered = 1;
    }
    return (int)r->n_reg++;
}

ACCESSOR(ovhip_lmcs_region, ovhip_rec_lmcs_regions, reg, n_reg)

/* tiny[k][0 .. 3]: how many commands at the END of class k are "tiny" -- plain 8x8, 4x8, 8x4, 4x4 blocks (in that order in the list:
 * OVHIP_TB_TR without LFNST off the arena's sub-blocks, or OVHIP_TB_DC), which k_itx_all takes a lane per sample, 4 / 8 / 8 / 16 blocks
 * to a workgroup, with the command as per-lane data (over half of a 4K picture's blocks).  Library-internal (the picture job);
 * ovhip_rec_tb_cmds_split is the same order. */
/* the same split of the commands [first, first + n) into `out` (n entries; the picture job's staging block of a band) */
void
ovhip_rec_tb_split_range_(const ovhip_recorder *r, size_t first, size_t n, ovhip_tb_cmd *out, size_t counts[4], size_t tiny[4][4])
{
    size_t start[20], cnt[20], k;
    const ovhip_tb_cmd *tb = r->tb + first;
    memset(cnt, 0, sizeof(cnt)); memset(counts, 0, 4 * sizeof(size_t)); memset(tiny, 0, 16 * sizeof(size_t));
    /* "small" = what one wavefront and a 4 KB slice of LDS take: at most 256 samples, no side above 32 */
#define TB_CLASS(c) (((c)->plane != 0) * 2 + ((c)->log2_w + (c)->log2_h <= 8 && (c)->log2_w <= 5 && (c)->log2_h <= 5))
    /* 0: a wave of its own through the general body; 1: 8x8, 2: 4x8, 3: 8x4, 4: 4x4 */
#define TB_SHAPE(c) (((c)->kind == OVHIP_TB_DC || ((c)->kind == OVHIP_TB_TR && !((c)->lfnst & 1))) ? \
                     ((c)->log2_w == 3 && (c)->log2_h == 3 ? 1 : (c)->log2_w == 2 && (c)->log2_h == 3 ? 2 : \
                      (c)->log2_w == 3 && (c)->log2_h == 2 ? 3 : (c)->log2_w == 2 && (c)->log2_h == 2 ? 4 : 0) : 0)
    for (size_t i = 0; i < n; ++i) cnt[5 * TB_CLASS(&tb[i]) + TB_SHAPE(&tb[i])]++;
    for (k = 0, start[0] = 0; k < 19; ++k) start[k + 1] = start[k] + cnt[k];
    for (size_t i = 0; i < n; ++i) out[start[5 * TB_CLASS(&tb[i]) + TB_SHAPE(&tb[i])]++] = tb[i];
#undef TB_CLASS
#undef TB_SHAPE
    for (k = 0; k < 4; ++k) {
        counts[k] = cnt[5 * k] + cnt[5 * k + 1] + cnt[5 * k + 2] + cnt[5 * k + 3] + cnt[5 * k + 4];
        for (int q = 0; q < 4; ++q) tiny[k][q] = cnt[5 * k + 1 + q];
    }
}

const ovhip_tb_cmd *
ovhip_rec_tb_cmds_split_tiny_(ovhip_recorder *r, size_t counts[4], size_t tiny[4][4], size_t *n)
{
    *n = r->n_tb;
    memset(counts, 0, 4 * sizeof(size_t)); memset(tiny, 0, 16 * sizeof(size_t));
    if (!r->n_tb) return r->tb;
    if (grow((void **)&r->tb_split, &r->cap_split, r->n_tb, sizeof(ovhip_tb_cmd))) { *n = 0; return NULL; }
    ovhip_rec_tb_split_range_(r, 0, r->n_tb, r->tb_split, counts, tiny);
    return r->tb_split;
}

const ovhip_tb_cmd *
ovhip_rec_tb_cmds_split(ovhip_recorder *r, size_t counts[4], size_t *n)
{
    size_t tiny[4][4];
    return ovhip_rec_tb_cmds_split_tiny_(r, counts, tiny, n);
}

/* ---------------------------------------------------------------- CIIP blend (rcn_ciip_weighted_sum) */
int
ovhip_rec_ciip(ovhip_recorder *r, int32_t x0, int32_t y0, int32_t log2_w, int32_t log2_h, int32_t mode_abv, int32_t mode_lft)
{
    if (x0 < 0 || y0 < 0 || log2_w < 2 || log2_h < 2 || log2_w > 6 || log2_h > 6) return OVHIP_EINVAL;
    if (r->log) ovhip_calllog_ciip_(r->log, x0, y0, log2_w, log2_h, mode_abv, mode_lft);
    if (grow((void **)&r->ciip, &r->cap_ciip, r->n_ciip + 1, sizeof(*r->ciip))) return OVHIP_ENOMEM;
    ovhip_ciip_unit *u = &r->ciip[r->n_ciip++];
    /* OV_INTRA = 2, OV_MIP = 4 (cu_utils.h:132-139) */
    const int intra_abv = mode_abv == 2 || mode_abv == 4, intra_lft = mode_lft == 2 || mode_lft == 4;
    u->x = (uint16_t)x0; u->y = (uint16_t)y0;
    u->log2_w = (uint8_t)log2_w; u->log2_h = (uint8_t)log2_h;
    u->wt = (uint8_t)(1 + intra_abv + intra_lft);
    u->chroma_inter = log2_w <= 2;
    return 1;
}

int
ovhip_ciip_weight(int32_t mode_abv, int32_t mode_lft)
{
    /* OV_INTRA = 2, OV_MIP = 4 (cu_utils.h:132-139) */
    return 1 + (mode_abv == 2 || mode_abv == 4) + (mode_lft == 2 || mode_lft == 4);
}

ACCESSOR(ovhip_ciip_unit, ovhip_rec_ciip_units, ciip, n_ciip)
