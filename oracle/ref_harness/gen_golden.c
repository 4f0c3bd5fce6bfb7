/* gen_golden.c -- TEST INFRASTRUCTURE (this container only).
 *
 * Drives the COMPILED REFERENCE (oracle/_ref/libovvcref.so, built from /root/reference where it
 * lies) through its own orchestrator slots on seeded inputs and writes small golden fixtures
 * to tests/golden/*.ovg.  The inputs are stored in the vocabulary of include/ovvc_hip.h
 * (ovhip_tu_state / ovhip_tu_desc / ovhip_pu_desc = what the slots receive), the expected
 * outputs are the bytes the reference produced.  Nothing from the reference is copied; the
 * reference does not travel to the GPU box, the fixtures do.
 *
 *   itx.ovg : tmp.rcn_tu_st / tmp.rcn_tu_c  (rcn_transform_tree.c:1228-1382)  -> K1..K4
 *   mc.ovg  : rcn_mcp_b / rcn_mcp_b_l / rcn_mcp_b_c (rcn_inter.c:2769-2966)   -> K5, K6, K11
 */
#include "ref_common.h"
#include "ovvc_hip.h"
#include "shim_stream.h"
#include <time.h>

/* "time" mode: seconds spent inside the reference's slots per stage (profiles/cpu_calibration.json: the oracle port is timed on
 * the same cases from Python, tools/cpu_calibration.py) */
enum { TS_ITX, TS_MC, TS_MCX, TS_MCA, TS_DBF, TS_SAO, TS_ALF, TS_INTRA, TS_N };
static const char *g_ts_name[TS_N] = { "itx", "mc", "mcx", "mca", "dbf", "sao", "alf", "intra" };
static double g_ts[TS_N];
static int g_time;
static inline double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }
#define TIMED(k, stmt) do { if (g_time) { const double t0_ = now_s(); stmt; g_ts[k] += now_s() - t0_; } else { stmt; } } while (0)

/* struct TUInfo is private to rcn_transform_tree.c:51-66 (and duplicated in
 * vcl_transform_unit.c:47-75); the slot signature only forward-declares it. */
struct TBInfo { uint16_t last_pos; uint64_t sig_sb_map; };
struct TUInfo {
    uint8_t is_sbt; uint8_t cbf_mask; uint16_t pos_offset; uint8_t tr_skip_mask;
    uint8_t cu_mts_flag; uint8_t cu_mts_idx; uint8_t lfnst_flag; uint8_t lfnst_idx;
    struct TBInfo tb_info[3];
};

extern uint64_t residual_coding_dpq(OVCTUDec *const, int16_t *const, uint8_t, uint8_t, uint16_t);

static void stub_intra_pred_c(const struct OVRCNCtx *const r, uint8_t m, int x0, int y0, int lw, int lh, CUFlags f)
{ (void)r; (void)m; (void)x0; (void)y0; (void)lw; (void)lh; (void)f; }

static int16_t rnd_coef(void)
{
    int k = rnd_range(0, 15);
    if (k < 9)  return (int16_t)rnd_range(-24, 24);
    if (k < 13) return (int16_t)rnd_range(-900, 900);
    if (k < 15) return (int16_t)rnd_range(-32768, 32767);
    return (int16_t)(rnd_range(0, 1) ? 32767 : -32768);
}

/* fill one TB worth of coefficients in the reference layout; returns number of int16 used */
static int
make_coefs(int16_t *dst, int log2_w, int log2_h, int pattern, int raster, uint64_t *map, uint16_t *last_pos)
{
    int w = 1 << log2_w, h = 1 << log2_h;
    if (raster) {
        int n = w * h;
        memset(dst, 0, n * 2);
        if (pattern == 0) { dst[0] = rnd_coef(); *last_pos = 0; }
        else { for (int i = 0; i < n; ++i) if (pattern == 2 || rnd_range(0, 2) == 0) dst[i] = rnd_coef(); *last_pos = 0x0101; }
        *map = 1;
        return n;
    }
    int cw = w > 32 ? 32 : w, ch = h > 32 ? 32 : h;
    int nx = cw / 4, ny = ch / 4;
    memset(dst, 0, cw * ch * 2);
    *map = 0;
    if (pattern == 0) {
        dst[0] = rnd_coef();
        if (!dst[0]) dst[0] = 7;
        *last_pos = 0;
        *map = rnd_range(0, 1);
        return cw * ch;
    }
    int lim_x = pattern == 2 ? nx : rnd_range(1, nx), lim_y = pattern == 2 ? ny : rnd_range(1, ny);
    for (int sy = 0; sy < lim_y; ++sy) {
        for (int sx = 0; sx < lim_x; ++sx) {
            if (pattern != 2 && (sx || sy) && rnd_range(0, 2) == 0) continue;
            *map |= 1ull << (sy * 8 + sx);
            int16_t *sb = dst + sy * 4 * cw + sx * 16;
            for (int i = 0; i < 16; ++i) sb[i] = (pattern == 2 || rnd_range(0, 1)) ? rnd_coef() : 0;
        }
    }
    *last_pos = 0x0302;
    return cw * ch;
}

static void
dump_rect(gbuf *exp, const uint16_t *p, int stride, int x, int y, int w, int h)
{
    for (int j = 0; j < h; ++j) gbuf_push(exp, p + (y + j) * stride + x, w);
}

/* derive_lfnst_mode_c (drv_lfnst.c:94-121) reads CTU-local luma mode maps; the shim performs
 * this lookup at record time.  For the fixtures the chroma mode is an explicit angular mode. */
static int shim_lfnst_mode_c(int log2_w, int log2_h, int mode)
{
    static const uint8_t ms_lut[6] = { 0, 6, 10, 12, 14, 15 };
    if (mode > 1) {
        int d = log2_w - log2_h; int ms = ms_lut[d < 0 ? -d : d];
        if (log2_w > log2_h && mode < 2 + ms) mode += 65;
        else if (log2_h > log2_w && mode > 66 - ms) mode -= 67;
    }
    return mode < 0 ? mode + 14 + 67 : mode >= 67 ? mode + 14 : mode;
}

/* TUInfo slots tmp.rcn_transform_tree visits for a CU (index arithmetic of rcn_transform_tree.c:1454-1506) */
static void
tt_slots(int l2w, int l2h, int max_tb, int depth, int base, int *out, int *n)
{
    int sv = l2w > max_tb, sh = l2h > max_tb, nsub = depth ? 1 : (1 << (sv + sh));
    if (l2w > 6 && l2h < 7) { tt_slots(6, l2h, max_tb, depth + 1, base, out, n); tt_slots(6, l2h, max_tb, depth + 1, base + 8, out, n); return; }
    if (l2h > 6 && l2w < 7) { tt_slots(l2w, 6, max_tb, depth + 1, base, out, n); tt_slots(l2w, 6, max_tb, depth + 1, base + 8, out, n); return; }
    if (sv || sh) {
        tt_slots(l2w - sv, l2h - sh, max_tb, depth + 1, base, out, n);
        if (sv) tt_slots(l2w - sv, l2h - sh, max_tb, depth + 1, base + nsub, out, n);
        if (sh) tt_slots(l2w - sv, l2h - sh, max_tb, depth + 1, base + 2 * nsub, out, n);
        if (sv && sh) tt_slots(l2w - sv, l2h - sh, max_tb, depth + 1, base + 3 * nsub, out, n);
        return;
    }
    out[(*n)++] = base;
}

static void
gen_itx(const char *dir)
{
    static uint16_t pred_y[128 * 128], pred_cb[64 * 64], pred_cr[64 * 64];
    gbuf b_state = { .type = T_U8 }, b_desc = { .type = T_U8 }, b_coff = { .type = T_U32 }, b_clen = { .type = T_U32 };
    gbuf b_coefs = { .type = T_I16 }, b_eoff = { .type = T_U32 }, b_exp = { .type = T_U16 };
    uint32_t n_cases = 0;
    g_seed = 0x266;

    fill_plane(pred_y, 128, 128, 128);
    fill_plane(pred_cb, 64, 64, 64);
    fill_plane(pred_cr, 64, 64, 64);
    /* make clipping at both ends reachable */
    for (int i = 0; i < 128 * 128; i += 37) pred_y[i] = (i & 1) ? 1023 : 0;
    for (int i = 0; i < 64 * 64; i += 29) { pred_cb[i] = (i & 1) ? 1023 : 0; pred_cr[i] = (i & 1) ? 0 : 1023; }

    OVCTUDec *c = ref_new_ctudec(0, 0);
    c->rcn_funcs.intra_pred_c = stub_intra_pred_c;
    const struct OVBuffInfo *cb = &c->rcn_ctx.ctu_buff;
    struct shim_stream S, ST;
    shim_stream_init(&S); shim_stream_init(&ST);
    if (g_shim) shim_bind(c, 128, 128, 0, 0);

    for (int tree = 0; tree <= 2; tree += 2) {
        int lmin = tree ? 1 : 2, lmax = tree ? 5 : 6;
        for (int l2w = lmin; l2w <= lmax; ++l2w) {
            for (int l2h = lmin; l2h <= lmax; ++l2h) {
                int reps = tree ? 5 : 12;
                for (int rep = 0; rep < reps; ++rep) {
                    ovhip_tu_state st; ovhip_tu_desc d; struct TUInfo tu;
                    memset(&st, 0, sizeof(st)); memset(&d, 0, sizeof(d)); memset(&tu, 0, sizeof(tu));
                    int w = 1 << l2w, h = 1 << l2h;
                    int unit = tree ? 2 : 4;
                    int x0 = rnd_range(0, ((tree ? 64 : 128) - w) / unit) * unit;
                    int y0 = rnd_range(0, ((tree ? 64 : 128) - h) / unit) * unit;
                    /* chroma TB size seen by the chroma path */
                    int cl2w = tree ? l2w : l2w - 1, cl2h = tree ? l2h : l2h - 1;

                    st.qp_y = rnd_range(0, 75); st.qp_cb = rnd_range(0, 75); st.qp_cr = rnd_range(0, 75); st.qp_jcbcr = rnd_range(0, 75);
                    st.qp_y_skip = st.qp_y < 16 ? 16 : st.qp_y; st.qp_cb_skip = st.qp_cb < 16 ? 16 : st.qp_cb;
                    st.qp_cr_skip = st.qp_cr < 16 ? 16 : st.qp_cr; st.qp_jcbcr_skip = st.qp_jcbcr < 16 ? 16 : st.qp_jcbcr;
                    st.dep_quant = rnd_range(0, 1);
                    st.mts_implicit = rnd_range(0, 1);
                    st.sh_ts_disabled = rnd_range(0, 1);
                    st.ict_type = rnd_range(0, 3);
                    st.lmcs_scale_c = rnd_range(0, 1);
                    st.lmcs_chroma_scale = (int16_t)rnd_range(1200, 3400);
                    st.intra_mode = (int8_t)rnd_range(0, 66);

                    d.x0 = x0; d.y0 = y0; d.log2_tb_w = l2w; d.log2_tb_h = l2h; d.tree = tree;

                    int variant = rep % 6;   /* 0 inter dct2, 1 implicit MTS intra, 2 explicit MTS, 3 LFNST, 4 TS, 5 inter */
                    int max_l2 = l2w > l2h ? l2w : l2h;
                    if (variant == 2 && (max_l2 > 5 || tree)) variant = 0;
                    if (variant == 4 && max_l2 > 5) variant = 5;
                    if (variant == 3 && tree && (l2w < 2 || l2h < 2)) variant = 0;
                    if (variant == 1 || variant == 3) d.cu_flags |= 1u << 1;                  /* pred_mode_flag (intra) */
                    if (variant == 2) { d.cu_mts_flag = 1; d.cu_mts_idx = rnd_range(0, 3); d.cu_flags |= (rnd_range(0, 1) << 1); }
                    if (variant == 3) { d.lfnst_flag = 1; d.lfnst_idx = rnd_range(0, 1); }

                    static const uint8_t cbf_st[8] = { 0x10, 0x12, 0x11, 0x13, 0x1b, 0x1a, 0x19, 0x0b };
                    static const uint8_t cbf_c[6] = { 0x2, 0x1, 0x3, 0xb, 0xa, 0x9 };
                    d.cbf_mask = tree ? cbf_c[rnd_range(0, 5)] : cbf_st[rnd_range(0, 7)];
                    if (!tree && l2w + l2h < 6) d.cbf_mask &= 0x10;          /* no 2x2 / 2x4 chroma TBs in single tree */
                    if (!d.cbf_mask) d.cbf_mask = 0x10;
                    if (variant == 4) {
                        d.tr_skip_mask = 0;
                        if (d.cbf_mask & 0x10) d.tr_skip_mask |= 0x10;
                        if (rnd_range(0, 1)) d.tr_skip_mask |= 0x3;
                        if (rep >= 6 || (tree && rnd_range(0, 1))) {
                            /* block DPCM (intra CUs): transform skip is implied, direction random */
                            d.cu_flags |= 1u << 1;
                            if (d.tr_skip_mask & 0x10) d.cu_flags |= (1u << 8) | ((uint32_t)rnd_range(0, 1) << 10);
                            if (d.tr_skip_mask & 0x3)  d.cu_flags |= (1u << 9) | ((uint32_t)rnd_range(0, 1) << 11);
                        }
                    }

                    /* chroma LFNST only exists in the dual-tree chroma path */
                    int c_mode = rnd_range(0, 66);
                    st.lfnst_mode_c = (int8_t)shim_lfnst_mode_c(cl2w, cl2h, c_mode);

                    /* ---- coefficients ---- */
                    int16_t *res[3] = { c->residual_cb, c->residual_cr, c->residual_y };
                    uint32_t off3[3] = { 0, 0, 0 }, len3[3] = { 0, 0, 0 };
                    int pos_offset = rnd_range(0, 3) * 16;
                    for (int comp = 0; comp < 3; ++comp) {
                        int is_l = comp == 2;
                        int used = is_l ? (d.cbf_mask & 0x10) && tree != 2
                                        : (tree != 1) && ((d.cbf_mask & 0x8) ? comp == 0 : (d.cbf_mask & (comp ? 0x1 : 0x2)));
                        if (!used) continue;
                        int tl2w = is_l ? l2w : cl2w, tl2h = is_l ? l2h : cl2h;
                        int ts = is_l ? !!(d.tr_skip_mask & 0x10)
                                      : (d.cbf_mask & 0x8) ? !!(d.tr_skip_mask & 0x1) : !!(d.tr_skip_mask & (comp ? 0x1 : 0x2));
                        int raster = (ts && !st.sh_ts_disabled) || tl2w < 2 || tl2h < 2;
                        int pattern = variant == 3 ? 1 : rnd_range(0, 2);
                        uint64_t map; uint16_t lp;
                        int n = make_coefs(res[comp] + pos_offset, tl2w, tl2h, pattern, raster, &map, &lp);
                        if (variant == 3) { map = 1; lp = 0x0101; }   /* LFNST: first sub-block only */
                        if (variant == 3 && !raster) {
                            int cw = (1 << tl2w) > 32 ? 32 : (1 << tl2w), ch = (1 << tl2h) > 32 ? 32 : (1 << tl2h);
                            memset(res[comp] + pos_offset + 16, 0, (cw * ch - 16) * 2);
                        }
                        d.sig_sb_map[comp] = map; d.last_pos[comp] = lp;
                        tu.tb_info[comp].sig_sb_map = map; tu.tb_info[comp].last_pos = lp;
                        off3[comp] = (uint32_t)b_coefs.n; len3[comp] = (uint32_t)n;
                        gbuf_push(&b_coefs, res[comp] + pos_offset, n);
                    }

                    /* ---- reference state ---- */
                    rcn_init_ict_functions_10(&c->rcn_funcs, st.ict_type, 10);
                    if (g_shim) rcn_init_functions_hip(&c->rcn_funcs, st.ict_type, 1, 0, 0, 10);   /* per slice in the decoder (slicedec.c:1462) */
                    c->dequant_luma.qp = st.qp_y; c->dequant_cb.qp = st.qp_cb; c->dequant_cr.qp = st.qp_cr;
                    c->dequant_joint_cb_cr.qp = st.qp_jcbcr;
                    c->dequant_luma_skip.qp = st.qp_y_skip; c->dequant_cb_skip.qp = st.qp_cb_skip;
                    c->dequant_cr_skip.qp = st.qp_cr_skip; c->dequant_jcbcr_skip.qp = st.qp_jcbcr_skip;
                    c->residual_coding_l = st.dep_quant ? &residual_coding_dpq : NULL;
                    c->mts_implicit = st.mts_implicit;
                    c->sh_ts_disabled = st.sh_ts_disabled;
                    c->lmcs_info.scale_c_flag = st.lmcs_scale_c;
                    c->lmcs_info.lmcs_chroma_scale = (uint16_t)st.lmcs_chroma_scale;
                    c->intra_mode = (uint8_t)st.intra_mode;
                    c->intra_mode_c = (uint8_t)c_mode;
                    c->qp_ctx.qp_bd_offset = 12;
                    tu.cbf_mask = d.cbf_mask; tu.pos_offset = (uint16_t)pos_offset; tu.tr_skip_mask = d.tr_skip_mask;
                    tu.cu_mts_flag = d.cu_mts_flag; tu.cu_mts_idx = d.cu_mts_idx;
                    tu.lfnst_flag = d.lfnst_flag; tu.lfnst_idx = d.lfnst_idx;

                    for (int j = 0; j < 128; ++j) memcpy(cb->y + j * cb->stride, pred_y + j * 128, 256);
                    for (int j = 0; j < 64; ++j) {
                        memcpy(cb->cb + j * cb->stride_c, pred_cb + j * 64, 128);
                        memcpy(cb->cr + j * cb->stride_c, pred_cr + j * 64, 128);
                    }
                    memset(&c->dbf_info, 0, sizeof(c->dbf_info));

                    if (g_shim && tree == 2 && d.lfnst_flag) {
                        /* derive_lfnst_mode_c's inputs as the decoder holds them: an explicit angular chroma mode */
                        c->part_ctx_c = &g_part;
                    }
                    if (tree == 0) TIMED(TS_ITX, c->rcn_funcs.tmp.rcn_tu_st(c, x0, y0, l2w, l2h, d.cu_flags, d.cbf_mask, &tu));
                    else           TIMED(TS_ITX, c->rcn_funcs.tmp.rcn_tu_c(c, x0, y0, l2w, l2h, d.cu_flags, d.cbf_mask, &tu));
                    if (g_shim) shim_case_end(c, &S, "itx");

                    uint32_t eoff[3] = { 0, 0, 0 };
                    if (tree == 0) {
                        eoff[0] = (uint32_t)b_exp.n; dump_rect(&b_exp, cb->y, cb->stride, x0, y0, w, h);
                        eoff[1] = (uint32_t)b_exp.n; dump_rect(&b_exp, cb->cb, cb->stride_c, x0 >> 1, y0 >> 1, w >> 1, h >> 1);
                        eoff[2] = (uint32_t)b_exp.n; dump_rect(&b_exp, cb->cr, cb->stride_c, x0 >> 1, y0 >> 1, w >> 1, h >> 1);
                    } else {
                        eoff[1] = (uint32_t)b_exp.n; dump_rect(&b_exp, cb->cb, cb->stride_c, x0, y0, w, h);
                        eoff[2] = (uint32_t)b_exp.n; dump_rect(&b_exp, cb->cr, cb->stride_c, x0, y0, w, h);
                    }
                    gbuf_push(&b_state, &st, sizeof(st));
                    gbuf_push(&b_desc, &d, sizeof(d));
                    gbuf_push(&b_coff, off3, 3); gbuf_push(&b_clen, len3, 3); gbuf_push(&b_eoff, eoff, 3);
                    n_cases++;
                }
            }
        }
    }

    /* ---- whole transform trees through tmp.rcn_transform_tree (rcn_transform_tree.c:1454-1506): inter CUs up to
     * 128x128, maximum transform size 64 or 32, one struct TUInfo per leaf as the caller lays them out ---- */
    gbuf t_state = { .type = T_U8 }, t_desc = { .type = T_U8 }, t_info = { .type = T_U8 }, t_coef = { .type = T_I16 };
    gbuf t_coff = { .type = T_U32 }, t_eoff = { .type = T_U32 };
    uint32_t n_tt = 0;
    {
        extern int transform_unit_st(OVCTUDec *const, unsigned int, unsigned int, unsigned int, unsigned int, uint8_t, CUFlags, uint8_t, struct TUInfo *const);
        c->transform_unit = (void *)&transform_unit_st;
        static const uint8_t shapes[][3] = { {6,6,6}, {7,6,6}, {6,7,6}, {7,7,6}, {7,5,6}, {5,7,6}, {6,6,5}, {6,5,5}, {7,7,5}, {5,5,6}, {7,4,6}, {6,6,6} };
        for (unsigned sc = 0; sc < sizeof(shapes) / 3; ++sc) {
            const int l2w = shapes[sc][0], l2h = shapes[sc][1], max_tb = shapes[sc][2];
            ovhip_tu_state st; ovhip_tt_desc td; ovhip_tu_info ti[16]; struct TUInfo tu[16];
            memset(&st, 0, sizeof(st)); memset(&td, 0, sizeof(td)); memset(ti, 0, sizeof(ti)); memset(tu, 0, sizeof(tu));
            st.qp_y = rnd_range(20, 60); st.qp_cb = rnd_range(20, 60); st.qp_cr = rnd_range(20, 60); st.qp_jcbcr = rnd_range(20, 60);
            st.qp_y_skip = st.qp_y; st.qp_cb_skip = st.qp_cb; st.qp_cr_skip = st.qp_cr; st.qp_jcbcr_skip = st.qp_jcbcr;
            st.dep_quant = rnd_range(0, 1); st.ict_type = rnd_range(0, 3); st.lmcs_scale_c = rnd_range(0, 1);
            st.lmcs_chroma_scale = (int16_t)rnd_range(1200, 3400);
            td.x0 = 0; td.y0 = 0; td.log2_w = l2w; td.log2_h = l2h; td.log2_max_tb_s = max_tb; td.tree = 0; td.cu_flags = 0;
            memset(c->residual_y, 0, sizeof(c->residual_y)); memset(c->residual_cb, 0, sizeof(c->residual_cb)); memset(c->residual_cr, 0, sizeof(c->residual_cr));
            /* leaf geometry: what the walker will hand out, in its own visiting order per TUInfo slot */
            const int lw = l2w > max_tb ? max_tb : l2w, lh = l2h > max_tb ? max_tb : l2h;
            uint32_t used = 0;
            int slots[64], n_slots = 0;
            tt_slots(l2w, l2h, max_tb, 0, 0, slots, &n_slots);
            const uint32_t leaf_sz = (uint32_t)(1 << ((lw > 5 ? 5 : lw) + (lh > 5 ? 5 : lh)));
            for (int q = 0; q < n_slots; ++q) {
                const int k = slots[q];
                if (k > 15 || ti[k].pos_offset || tu[k].cbf_mask) continue;      /* a slot may be visited twice (depth > 1): keep the first fill */
                static const uint8_t cbfs[6] = { 0x10, 0x13, 0x1b, 0x12, 0x00, 0x11 };
                ti[k].cbf_mask = cbfs[rnd_range(0, 5)];
                ti[k].pos_offset = (uint16_t)used;
                int16_t *res[3] = { c->residual_cb, c->residual_cr, c->residual_y };
                for (int comp = 0; comp < 3; ++comp) {
                    int is_l = comp == 2;
                    int use = is_l ? (ti[k].cbf_mask & 0x10) : ((ti[k].cbf_mask & 0x8) ? comp == 0 : (ti[k].cbf_mask & (comp ? 0x1 : 0x2)));
                    if (!use) continue;
                    uint64_t map; uint16_t lp;
                    make_coefs(res[comp] + ti[k].pos_offset, is_l ? lw : lw - 1, is_l ? lh : lh - 1, rnd_range(0, 2), 0, &map, &lp);
                    ti[k].sig_sb_map[comp] = map; ti[k].last_pos[comp] = lp;
                    tu[k].tb_info[comp].sig_sb_map = map; tu[k].tb_info[comp].last_pos = lp;
                }
                tu[k].cbf_mask = ti[k].cbf_mask; tu[k].pos_offset = ti[k].pos_offset;
                used += leaf_sz;
            }
            rcn_init_ict_functions_10(&c->rcn_funcs, st.ict_type, 10);
            if (g_shim) rcn_init_functions_hip(&c->rcn_funcs, st.ict_type, 1, 0, 0, 10);
            c->dequant_luma.qp = st.qp_y; c->dequant_cb.qp = st.qp_cb; c->dequant_cr.qp = st.qp_cr; c->dequant_joint_cb_cr.qp = st.qp_jcbcr;
            c->dequant_luma_skip.qp = st.qp_y_skip; c->dequant_cb_skip.qp = st.qp_cb_skip; c->dequant_cr_skip.qp = st.qp_cr_skip; c->dequant_jcbcr_skip.qp = st.qp_jcbcr_skip;
            c->residual_coding_l = st.dep_quant ? &residual_coding_dpq : NULL;
            c->mts_implicit = 0; c->sh_ts_disabled = 0; c->tmp_ciip = 0;
            c->lmcs_info.scale_c_flag = st.lmcs_scale_c; c->lmcs_info.lmcs_chroma_scale = (uint16_t)st.lmcs_chroma_scale;
            for (int j = 0; j < 128; ++j) memcpy(cb->y + j * cb->stride, pred_y + j * 128, 256);
            for (int j = 0; j < 64; ++j) { memcpy(cb->cb + j * cb->stride_c, pred_cb + j * 64, 128); memcpy(cb->cr + j * cb->stride_c, pred_cr + j * 64, 128); }
            memset(&c->dbf_info, 0, sizeof(c->dbf_info));
            c->rcn_funcs.tmp.rcn_transform_tree(c, 0, 0, l2w, l2h, max_tb, 0, 0, tu);
            if (g_shim) shim_case_end(c, &ST, "itx tree");

            uint32_t coff = (uint32_t)t_coef.n, eoff[3];
            gbuf_push(&t_coef, c->residual_cb, used); gbuf_push(&t_coef, c->residual_cr, used); gbuf_push(&t_coef, c->residual_y, used);
            uint32_t co[2] = { coff, used };
            eoff[0] = (uint32_t)b_exp.n; dump_rect(&b_exp, cb->y, cb->stride, 0, 0, 1 << l2w, 1 << l2h);
            eoff[1] = (uint32_t)b_exp.n; dump_rect(&b_exp, cb->cb, cb->stride_c, 0, 0, 1 << (l2w - 1), 1 << (l2h - 1));
            eoff[2] = (uint32_t)b_exp.n; dump_rect(&b_exp, cb->cr, cb->stride_c, 0, 0, 1 << (l2w - 1), 1 << (l2h - 1));
            gbuf_push(&t_state, &st, sizeof(st)); gbuf_push(&t_desc, &td, sizeof(td)); gbuf_push(&t_info, ti, sizeof(ti));
            gbuf_push(&t_coff, co, 2); gbuf_push(&t_eoff, eoff, 3);
            n_tt++;
        }
    }

    if (g_shim) {
        shim_stream_write(dir, "shim_itx.ovg", &S, c, NULL, 0);
        shim_stream_write(dir, "shim_itx_tree.ovg", &ST, c, NULL, 0);
        return;
    }
    gfile g = gfile_open(dir, "itx.ovg");
    uint32_t d2[2];
    d2[0] = n_tt; d2[1] = sizeof(ovhip_tu_state); gfile_array(&g, "tt_state", T_U8, t_state.data, 2, d2);
    d2[1] = sizeof(ovhip_tt_desc);  gfile_array(&g, "tt_desc", T_U8, t_desc.data, 2, d2);
    d2[1] = 16 * sizeof(ovhip_tu_info); gfile_array(&g, "tt_info", T_U8, t_info.data, 2, d2);
    d2[1] = 2; gfile_array(&g, "tt_coef_off", T_U32, t_coff.data, 2, d2);
    d2[1] = 3; gfile_array(&g, "tt_exp_off", T_U32, t_eoff.data, 2, d2);
    gfile_buf(&g, "tt_coefs", &t_coef);
    d2[0] = 128; d2[1] = 128; gfile_array(&g, "pred_y", T_U16, pred_y, 2, d2);
    d2[0] = 64; d2[1] = 64;   gfile_array(&g, "pred_cb", T_U16, pred_cb, 2, d2);
    gfile_array(&g, "pred_cr", T_U16, pred_cr, 2, d2);
    d2[0] = n_cases; d2[1] = sizeof(ovhip_tu_state); gfile_array(&g, "state", T_U8, b_state.data, 2, d2);
    d2[1] = sizeof(ovhip_tu_desc); gfile_array(&g, "desc", T_U8, b_desc.data, 2, d2);
    d2[1] = 3; gfile_array(&g, "coef_off", T_U32, b_coff.data, 2, d2);
    gfile_array(&g, "coef_len", T_U32, b_clen.data, 2, d2);
    gfile_array(&g, "exp_off", T_U32, b_eoff.data, 2, d2);
    gfile_buf(&g, "coefs", &b_coefs);
    gfile_buf(&g, "exp", &b_exp);
    gfile_close(&g);
    fprintf(stderr, "itx.ovg: %u cases, %zu coef int16, %zu expected samples\n", n_cases, b_coefs.n, b_exp.n);
}

/* ====================================================================================== MC */
#define MC_W 208
#define MC_H 120

static void
gen_mc(const char *dir)
{
    gbuf b_desc = { .type = T_U8 }, b_eoff = { .type = T_U32 }, b_exp = { .type = T_U16 };
    uint32_t n_cases = 0;
    g_seed = 0x266 + 1;

    OVCTUDec *c = ref_new_ctudec(0, 0);
    struct InterDRVCtx *ic = &c->drv_ctx.inter_ctx;
    OVPicture *ref[3];
    for (int i = 0; i < 3; ++i) {
        ref[i] = ref_new_picture(MC_W, MC_H, 8 * (i + 1));
        fill_plane(ref[i]->frame->data[0], MC_W, MC_H, MC_W);
        fill_plane(ref[i]->frame->data[1], MC_W / 2, MC_H / 2, MC_W / 2);
        fill_plane(ref[i]->frame->data[2], MC_W / 2, MC_H / 2, MC_W / 2);
    }
    /* rpl0 = {ref0, ref1}, rpl1 = {ref2, ref0}: rpl0[0] and rpl1[1] are the SAME picture (identical-motion path) */
    ic->rpl0[0] = ref[0]; ic->rpl0[1] = ref[1]; ic->rpl1[0] = ref[2]; ic->rpl1[1] = ref[0];
    static const uint8_t slot0[2] = { 0, 1 }, slot1[2] = { 2, 0 };
    for (int i = 0; i < 16; ++i) {
        ic->scale_fact_rpl0[i][0] = ic->scale_fact_rpl0[i][1] = 1 << RPR_SCALE_BITS;
        ic->scale_fact_rpl1[i][0] = ic->scale_fact_rpl1[i][1] = 1 << RPR_SCALE_BITS;
    }
    const struct OVBuffInfo *cb = &c->rcn_ctx.ctu_buff;
    struct shim_stream S;
    shim_stream_init(&S);
    if (g_shim) shim_bind(c, MC_W, MC_H, 0, 0);

    for (int l2w = 2; l2w <= 7; ++l2w) {
        for (int l2h = 2; l2h <= 7; ++l2h) {
            int w = 1 << l2w, h = 1 << l2h;
            if (w > MC_W || h > MC_H) continue;
            int reps = (w * h <= 256) ? 40 : (w * h <= 1024 ? 14 : 4);
            for (int rep = 0; rep < reps; ++rep) {
                ovhip_pu_desc d;
                memset(&d, 0, sizeof(d));
                /* position: anywhere on the 4-grid such that the PU is inside the picture and one CTU */
                int px, py;
                do {
                    px = rnd_range(0, (MC_W - w) / 4) * 4;
                    py = rnd_range(0, (MC_H - h) / 4) * 4;
                } while ((px >> 7) != ((px + w - 1) >> 7) || (py >> 7) != ((py + h - 1) >> 7));
                d.x0 = px; d.y0 = py; d.log2_w = l2w; d.log2_h = l2h;
                d.inter_dir = rnd_range(1, 3);
                d.ref_idx0 = rnd_range(0, 1); d.ref_idx1 = rnd_range(0, 1);
                int range = rep % 4 == 0 ? 4000 : (rep % 4 == 1 ? 64 : 600);   /* far outside / tiny / normal, 1/16 pel */
                d.mv0x = rnd_range(-range, range); d.mv0y = rnd_range(-range, range);
                d.mv1x = rnd_range(-range, range); d.mv1y = rnd_range(-range, range);
                if (rep % 7 == 3) { d.mv0x &= ~15; d.mv1y &= ~15; }            /* H-only / V-only / copy paths */
                if (rep % 7 == 5) { d.mv0x &= ~15; d.mv0y &= ~15; d.mv1x &= ~15; }
                if (rep % 9 == 4) { d.inter_dir = 3; d.ref_idx0 = 0; d.ref_idx1 = 1; d.mv1x = d.mv0x; d.mv1y = d.mv0y; } /* identical motion */
                d.prec_amvr_half = rep % 5 == 2;
                if (d.prec_amvr_half) { d.mv0x = (d.mv0x & ~15) | 8; d.mv1y = (d.mv1y & ~15) | 8; }
                if (w == 4 && h == 4) d.prec_amvr_half = 0;
                d.bcw_idx_plus1 = (rep % 3 == 1) ? rnd_range(1, 5) : 0;
                d.planes = rep % 11 == 7 ? 1 : (rep % 11 == 9 ? 2 : 3);
                d.poc0 = ic->rpl0[d.ref_idx0]->poc; d.poc1 = ic->rpl1[d.ref_idx1]->poc;
                d.ref0 = slot0[d.ref_idx0]; d.ref1 = slot1[d.ref_idx1];

                c->ctb_x = px >> 7; c->ctb_y = py >> 7;
                int x0 = px & 127, y0 = py & 127;
                ic->prec_amvr = d.prec_amvr_half ? MV_PRECISION_HALF : 0;
                OVMV mv0 = { .x = d.mv0x, .y = d.mv0y, .ref_idx = d.ref_idx0, .bcw_idx_plus1 = d.bcw_idx_plus1 };
                OVMV mv1 = { .x = d.mv1x, .y = d.mv1y, .ref_idx = d.ref_idx1, .bcw_idx_plus1 = d.bcw_idx_plus1 };
                for (int j = 0; j < 128; ++j) memset(cb->y + j * cb->stride, 0xAB, 256);
                for (int j = 0; j < 64; ++j) { memset(cb->cb + j * cb->stride_c, 0xAB, 128); memset(cb->cr + j * cb->stride_c, 0xAB, 128); }

                if (d.planes == 3)
                    TIMED(TS_MC, c->rcn_funcs.rcn_mcp_b(c, *cb, ic, c->part_ctx, mv0, mv1, x0, y0, l2w, l2h, d.inter_dir, d.ref_idx0, d.ref_idx1));
                else if (d.planes == 1)
                    TIMED(TS_MC, c->rcn_funcs.rcn_mcp_b_l(c, *cb, ic, c->part_ctx, mv0, mv1, x0, y0, l2w, l2h, d.inter_dir, d.ref_idx0, d.ref_idx1));
                else
                    TIMED(TS_MC, c->rcn_funcs.rcn_mcp_b_c(c, *cb, ic, c->part_ctx, mv0, mv1, x0, y0, l2w, l2h, d.inter_dir, d.ref_idx0, d.ref_idx1));
                if (g_shim) shim_case_end(c, &S, "mc");

                uint32_t eoff[3];
                eoff[0] = (uint32_t)b_exp.n; dump_rect(&b_exp, cb->y, cb->stride, x0, y0, w, h);
                eoff[1] = (uint32_t)b_exp.n; dump_rect(&b_exp, cb->cb, cb->stride_c, x0 >> 1, y0 >> 1, w >> 1, h >> 1);
                eoff[2] = (uint32_t)b_exp.n; dump_rect(&b_exp, cb->cr, cb->stride_c, x0 >> 1, y0 >> 1, w >> 1, h >> 1);
                gbuf_push(&b_desc, &d, sizeof(d));
                gbuf_push(&b_eoff, eoff, 3);
                n_cases++;
            }
        }
    }

    if (g_shim) { shim_stream_write(dir, "shim_mc.ovg", &S, c, ref, 3); return; }
    gfile g = gfile_open(dir, "mc.ovg");
    uint32_t d3[3] = { 3, MC_H, MC_W };
    uint16_t *all = malloc(3 * MC_W * MC_H * 2);
    for (int i = 0; i < 3; ++i) memcpy(all + i * MC_W * MC_H, ref[i]->frame->data[0], MC_W * MC_H * 2);
    gfile_array(&g, "ref_y", T_U16, all, 3, d3);
    d3[1] = MC_H / 2; d3[2] = MC_W / 2;
    for (int i = 0; i < 3; ++i) memcpy(all + i * (MC_W / 2) * (MC_H / 2), ref[i]->frame->data[1], (MC_W / 2) * (MC_H / 2) * 2);
    gfile_array(&g, "ref_cb", T_U16, all, 3, d3);
    for (int i = 0; i < 3; ++i) memcpy(all + i * (MC_W / 2) * (MC_H / 2), ref[i]->frame->data[2], (MC_W / 2) * (MC_H / 2) * 2);
    gfile_array(&g, "ref_cr", T_U16, all, 3, d3);
    uint32_t d2[2] = { n_cases, sizeof(ovhip_pu_desc) };
    gfile_array(&g, "desc", T_U8, b_desc.data, 2, d2);
    d2[1] = 3; gfile_array(&g, "exp_off", T_U32, b_eoff.data, 2, d2);
    gfile_buf(&g, "exp", &b_exp);
    gfile_close(&g);
    fprintf(stderr, "mc.ovg: %u cases, %zu expected samples\n", n_cases, b_exp.n);
}


/* ====================================================================================== MCX
 * mcx.ovg : rcn_bdof_mcp_l (+ rcn_mcp_b_c) and rcn_dmvr_mv_refine driven the way their caller does
 *           (vcl_coding_unit.c:2450-2472, :2598-2668)                          -> K7, K8
 * Reference pictures: R0 smooth random; R1 = R0 displaced by a per-64x64-region integer shift plus a
 * per-region noise level (0 = exact copy: exercises DMVR's early exit and the BDOF disable);
 * R2 independent.  rpl0 = {R0, R2}, rpl1 = {R1, R0}. */
static void
gen_mcx(const char *dir)
{
    gbuf b_desc = { .type = T_U8 }, b_eoff = { .type = T_U32 }, b_exp = { .type = T_U16 }, b_mv = { .type = T_I32 };
    uint32_t n_cases = 0;
    g_seed = 0x266 + 77;

    OVCTUDec *c = ref_new_ctudec(0, 0);
    struct InterDRVCtx *ic = &c->drv_ctx.inter_ctx;
    OVPicture *ref[3];
    for (int i = 0; i < 3; ++i) {
        ref[i] = ref_new_picture(MC_W, MC_H, i == 0 ? 8 : (i == 1 ? 24 : 4));
        for (int p = 0; p < 3; ++p) fill_plane(ref[i]->frame->data[p], MC_W >> !!p, MC_H >> !!p, MC_W >> !!p);
    }
    for (int p = 0; p < 3; ++p) {
        int w = MC_W >> !!p, h = MC_H >> !!p, rs = p ? 32 : 64;
        uint16_t *a = (uint16_t *)ref[0]->frame->data[p], *b = (uint16_t *)ref[1]->frame->data[p];
        static int shx[16], shy[16], nz[16];
        if (!p) for (int k = 0; k < 16; ++k) { shx[k] = rnd_range(-2, 2); shy[k] = rnd_range(-2, 2); nz[k] = k % 3 == 0 ? 0 : (k % 3 == 1 ? 2 : 12); }
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                int k = ((y / rs) * 4 + x / rs) & 15;
                int sx = x + (p ? shx[k] / 2 : shx[k]), sy = y + (p ? shy[k] / 2 : shy[k]);
                sx = sx < 0 ? 0 : sx >= w ? w - 1 : sx; sy = sy < 0 ? 0 : sy >= h ? h - 1 : sy;
                int v = a[sy * w + sx] + (nz[k] ? rnd_range(-nz[k], nz[k]) : 0);
                b[y * w + x] = (uint16_t)(v < 0 ? 0 : v > 1023 ? 1023 : v);
            }
    }
    ic->rpl0[0] = ref[0]; ic->rpl0[1] = ref[2]; ic->rpl1[0] = ref[1]; ic->rpl1[1] = ref[0];
    static const uint8_t slot0[2] = { 0, 2 }, slot1[2] = { 1, 0 };
    for (int i = 0; i < 16; ++i) {
        ic->scale_fact_rpl0[i][0] = ic->scale_fact_rpl0[i][1] = 1 << RPR_SCALE_BITS;
        ic->scale_fact_rpl1[i][0] = ic->scale_fact_rpl1[i][1] = 1 << RPR_SCALE_BITS;
    }
    const struct OVBuffInfo *cb = &c->rcn_ctx.ctu_buff;
    struct shim_stream S;
    shim_stream_init(&S);
    /* shim mode: a picture-level TMVP motion plane per list, as ovdpb_init_picture allocates them (mvpool.c), so that the
     * write-back of the refined vectors can be checked against where the reference's caller + tmvp_store_mv put them */
    struct MVPlane pl0, pl1;
    const int nb_ctb_w = (MC_W + 127) / 128, nb_ctb_h = (MC_H + 127) / 128, pln_stride = 16 * nb_ctb_w;
    if (g_shim) {
        shim_bind(c, MC_W, MC_H, 0, 0);
        memset(&pl0, 0, sizeof(pl0)); memset(&pl1, 0, sizeof(pl1));
        pl0.mvs = calloc((size_t)pln_stride * 16 * nb_ctb_h, sizeof(OVMV)); pl1.mvs = calloc((size_t)pln_stride * 16 * nb_ctb_h, sizeof(OVMV));
        ic->tmvp_ctx.plane0 = &pl0; ic->tmvp_ctx.plane1 = &pl1;
        c->nb_ctb_pic_w = nb_ctb_w;
    }

    for (int l2w = 3; l2w <= 6; ++l2w) {
        for (int l2h = 3; l2h <= 6; ++l2h) {
            int w = 1 << l2w, h = 1 << l2h;
            if (l2w + l2h < 7 || w > MC_W || h > MC_H) continue;
            int reps = (w * h <= 256) ? 60 : (w * h <= 1024 ? 18 : 8);
            for (int rep = 0; rep < reps; ++rep) {
                ovhip_pu_desc d;
                memset(&d, 0, sizeof(d));
                int px, py;
                do {
                    px = rnd_range(0, (MC_W - w) / 4) * 4;
                    py = rnd_range(0, (MC_H - h) / 4) * 4;
                } while ((px >> 7) != ((px + w - 1) >> 7) || (py >> 7) != ((py + h - 1) >> 7));
                if (rep % 10 == 6) px = 0;
                if (rep % 10 == 8) py = 0;
                d.x0 = px; d.y0 = py; d.log2_w = l2w; d.log2_h = l2h;
                d.inter_dir = 3;
                d.ref_idx0 = rep % 6 == 5; d.ref_idx1 = rep % 8 == 7;
                int range = rep % 9 == 0 ? 4000 : (rep % 3 == 1 ? 40 : 300);
                d.mv0x = rnd_range(-range, range); d.mv0y = rnd_range(-range, range);
                if (rep % 4 != 3) { d.mv1x = -d.mv0x + rnd_range(-20, 20); d.mv1y = -d.mv0y + rnd_range(-20, 20); }   /* mirrored + jitter */
                else              { d.mv1x = rnd_range(-range, range); d.mv1y = rnd_range(-range, range); }
                if (rep % 7 == 3) { d.mv0x &= ~15; d.mv1y &= ~15; }
                if (rep % 7 == 5) { d.mv0x &= ~15; d.mv0y &= ~15; d.mv1x &= ~15; d.mv1y &= ~15; }
                d.prec_amvr_half = rep % 5 == 2;
                if (d.prec_amvr_half) { d.mv0x = (d.mv0x & ~15) | 8; d.mv1y = (d.mv1y & ~15) | 8; }
                d.planes = 3;
                d.refine = (uint8_t)(1 + rep % 3);                  /* BDOF, DMVR, DMVR + BDOF */
                d.poc0 = ic->rpl0[d.ref_idx0]->poc; d.poc1 = ic->rpl1[d.ref_idx1]->poc;
                d.ref0 = slot0[d.ref_idx0]; d.ref1 = slot1[d.ref_idx1];

                c->ctb_x = px >> 7; c->ctb_y = py >> 7;
                int x0 = px & 127, y0 = py & 127;
                ic->prec_amvr = d.prec_amvr_half ? MV_PRECISION_HALF : 0;
                OVMV mv0 = { .x = d.mv0x, .y = d.mv0y, .ref_idx = d.ref_idx0 };
                OVMV mv1 = { .x = d.mv1x, .y = d.mv1y, .ref_idx = d.ref_idx1 };
                for (int j = 0; j < 128; ++j) memset(cb->y + j * cb->stride, 0xAB, 256);
                for (int j = 0; j < 64; ++j) { memset(cb->cb + j * cb->stride_c, 0xAB, 128); memset(cb->cr + j * cb->stride_c, 0xAB, 128); }

                int l2sw = l2w < 4 ? l2w : 4, l2sh = l2h < 4 ? l2h : 4;
                uint32_t mv_off = (uint32_t)b_mv.n;
                for (int i = 0; i < (h >> l2sh); ++i)
                    for (int j = 0; j < (w >> l2sw); ++j) {
                        OVMV m0 = mv0, m1 = mv1;
                        if (d.refine & OVHIP_PU_DMVR)
                            c->rcn_funcs.rcn_dmvr_mv_refine(c, *cb, x0 + j * 16, y0 + i * 16, l2sw, l2sh, &m0, &m1,
                                                            d.ref_idx0, d.ref_idx1, d.refine & OVHIP_PU_BDOF);
                        else
                            c->rcn_funcs.rcn_bdof_mcp_l(c, *cb, x0 + j * 16, y0 + i * 16, l2sw, l2sh, m0, m1, d.ref_idx0, d.ref_idx1);
                        int32_t o[4] = { m0.x, m0.y, m1.x, m1.y };
                        gbuf_push(&b_mv, o, 4);
                    }
                if (!(d.refine & OVHIP_PU_DMVR))
                    c->rcn_funcs.rcn_mcp_b_c(c, *cb, ic, c->part_ctx, mv0, mv1, x0, y0, l2w, l2h, 3, d.ref_idx0, d.ref_idx1);
                if (g_shim) {
                    if (d.refine & OVHIP_PU_DMVR) {
                        /* vectors "from the device" for this case's units: a recognisable value per unit and component;
                         * then every entry the reference's flow would have written must hold it, and no other entry */
                        const int nu = (h >> l2sh) * (w >> l2sw);
                        int32_t fake[64 * 4];
                        for (int u = 0; u < nu; ++u) for (int k = 0; k < 4; ++k) fake[4 * u + k] = 100000 + (int32_t)n_cases * 1000 + u * 8 + k;
                        memset(pl0.mvs, 0, (size_t)pln_stride * 16 * nb_ctb_h * sizeof(OVMV)); memset(pl1.mvs, 0, (size_t)pln_stride * 16 * nb_ctb_h * sizeof(OVMV));
                        {
                            /* the plane entries as the device derives them (k_tmvp_cells; its CPU restatement here: the harness runs
                             * without a GPU, tests/test_gpu_shim_replay.py holds the kernel to the same entries) */
                            extern void oracle_tmvp_cells(const ovhip_mc_unit *, uint32_t, const int32_t *, int, int, ovhip_tmvp_cell *);
                            size_t n_rec = 0;
                            const ovhip_mc_unit *ru = ovhip_rec_mcx_units(ovhip_shim_recorder(c), &n_rec);
                            if ((int)n_rec != nu) { fprintf(stderr, "shim: %zu refined units recorded, %d expected (case %u)\n", n_rec, nu, n_cases); exit(1); }
                            static ovhip_tmvp_cell cells[64 * 4];
                            oracle_tmvp_cells(ru, (uint32_t)nu, fake, 7, pln_stride / 16, cells);
                            ovhip_shim_apply_tmvp_cells(c, cells, 4 * (size_t)nu);
                        }
                        int32_t checked = 0, bad = 0;
                        static OVMV exp0[16 * 16], exp1[16 * 16];      /* the CTU-local tmvp_mv[] arrays of the caller */
                        memset(exp0, 0, sizeof(exp0)); memset(exp1, 0, sizeof(exp1));
                        for (int i = 0, u = 0; i < (h >> l2sh); ++i)
                            for (int j = 0; j < (w >> l2sw); ++j, ++u) {
                                /* vcl_coding_unit.c:2629-2645 */
                                const int b = ((x0 + 7 + j * 16) >> 3) + ((y0 + 7 + i * 16) >> 3) * 16;
                                for (int dy = 0; dy <= (l2sh > 3); ++dy) for (int dx = 0; dx <= (l2sw > 3); ++dx) {
                                    if (((x0 + 7 + j * 16) >> 3) + dx > 15 || ((y0 + 7 + i * 16) >> 3) + dy > 15) continue;
                                    exp0[b + dx + 16 * dy].x = fake[4 * u]; exp0[b + dx + 16 * dy].y = fake[4 * u + 1];
                                    exp1[b + dx + 16 * dy].x = fake[4 * u + 2]; exp1[b + dx + 16 * dy].y = fake[4 * u + 3];
                                }
                            }
                        /* tmvp_store_mv (drv_lines.c:270-330): row i of the CTU array -> plane->mvs + ctb_offset + i * pln_stride */
                        const int ctb_off = (c->ctb_x + c->ctb_y * pln_stride) * 16;
                        for (int py2 = 0; py2 < 16 * nb_ctb_h; ++py2)
                            for (int px2 = 0; px2 < pln_stride; ++px2) {
                                const int lx = px2 - c->ctb_x * 16, ly = py2 - c->ctb_y * 16;
                                OVMV e0 = { 0 }, e1 = { 0 };
                                if (lx >= 0 && lx < 16 && ly >= 0 && ly < 16) { e0 = exp0[lx + 16 * ly]; e1 = exp1[lx + 16 * ly]; }
                                const OVMV *g0 = &pl0.mvs[py2 * pln_stride + px2], *g1 = &pl1.mvs[py2 * pln_stride + px2];
                                (void)ctb_off;
                                if (g0->x != e0.x || g0->y != e0.y || g1->x != e1.x || g1->y != e1.y) bad++;
                                checked += e0.x != 0;
                                if (e0.x != 0) {
                                    /* which unit of the case the entry belongs to is encoded in the recognisable value */
                                    int32_t rec[3] = { (int32_t)S.n_cases, py2 * pln_stride + px2, (e0.x - 100000 - (int32_t)n_cases * 1000) / 8 };
                                    gbuf_push(&S.tmvp_exp, rec, 3);
                                }
                            }
                        if (bad) { fprintf(stderr, "shim: refined-MV write-back differs from the reference's flow in %d entries (case %u)\n", bad, n_cases); exit(1); }
                        gbuf_push(&S.mv_chk, &checked, 1);
                    }
                    shim_case_end(c, &S, "mcx");
                    /* next case: fresh patch list, as a new picture would */
                    extern void ovhip_shim_new_picture_for_test(OVCTUDec *);
                    ovhip_shim_new_picture_for_test(c);
                }

                uint32_t eoff[4];
                eoff[0] = (uint32_t)b_exp.n; dump_rect(&b_exp, cb->y, cb->stride, x0, y0, w, h);
                eoff[1] = (uint32_t)b_exp.n; dump_rect(&b_exp, cb->cb, cb->stride_c, x0 >> 1, y0 >> 1, w >> 1, h >> 1);
                eoff[2] = (uint32_t)b_exp.n; dump_rect(&b_exp, cb->cr, cb->stride_c, x0 >> 1, y0 >> 1, w >> 1, h >> 1);
                eoff[3] = mv_off;
                gbuf_push(&b_desc, &d, sizeof(d));
                gbuf_push(&b_eoff, eoff, 4);
                n_cases++;
            }
        }
    }

    if (g_shim) { shim_stream_write(dir, "shim_mcx.ovg", &S, c, ref, 3); return; }
    gfile g = gfile_open(dir, "mcx.ovg");
    uint32_t d3[3] = { 3, MC_H, MC_W };
    uint16_t *all = malloc(3 * MC_W * MC_H * 2);
    for (int i = 0; i < 3; ++i) memcpy(all + i * MC_W * MC_H, ref[i]->frame->data[0], MC_W * MC_H * 2);
    gfile_array(&g, "ref_y", T_U16, all, 3, d3);
    d3[1] = MC_H / 2; d3[2] = MC_W / 2;
    for (int i = 0; i < 3; ++i) memcpy(all + i * (MC_W / 2) * (MC_H / 2), ref[i]->frame->data[1], (MC_W / 2) * (MC_H / 2) * 2);
    gfile_array(&g, "ref_cb", T_U16, all, 3, d3);
    for (int i = 0; i < 3; ++i) memcpy(all + i * (MC_W / 2) * (MC_H / 2), ref[i]->frame->data[2], (MC_W / 2) * (MC_H / 2) * 2);
    gfile_array(&g, "ref_cr", T_U16, all, 3, d3);
    uint32_t d2[2] = { n_cases, sizeof(ovhip_pu_desc) };
    gfile_array(&g, "desc", T_U8, b_desc.data, 2, d2);
    d2[1] = 4; gfile_array(&g, "exp_off", T_U32, b_eoff.data, 2, d2);
    gfile_buf(&g, "exp", &b_exp);
    gfile_buf(&g, "exp_mv", &b_mv);
    gfile_close(&g);
    fprintf(stderr, "mcx.ovg: %u cases, %zu expected samples, %zu refined MVs\n", n_cases, b_exp.n, b_mv.n / 4);
}

/* ====================================================================================== MCA
 * mca.ovg : affine CUs the way rcn_affine_mcp_b_l / rcn_affine_prof_mcp_b_l / rcn_affine_mcp_b_c
 *           (drv_affine_mvp.c:3264-3411) drive rcn_mcp_b_l(2,2) / rcn_prof_mcp_b_l / rcn_mcp_b_c(3,3)  -> K9
 * struct PROFInfo is private to rcn_inter.c:1128-1134; the slot only forward-declares it. */
struct PROFInfo { int16_t dmv_scale_h_0[16], dmv_scale_v_0[16], dmv_scale_h_1[16], dmv_scale_v_1[16]; };

static void
gen_mca(const char *dir)
{
    gbuf b_desc = { .type = T_U8 }, b_eoff = { .type = T_U32 }, b_exp = { .type = T_U16 }, b_mv = { .type = T_I32 };
    uint32_t n_cases = 0;
    g_seed = 0x266 + 99;

    OVCTUDec *c = ref_new_ctudec(0, 0);
    struct InterDRVCtx *ic = &c->drv_ctx.inter_ctx;
    OVPicture *ref[3];
    for (int i = 0; i < 3; ++i) {
        ref[i] = ref_new_picture(MC_W, MC_H, 8 * (i + 1));
        for (int p = 0; p < 3; ++p) fill_plane(ref[i]->frame->data[p], MC_W >> !!p, MC_H >> !!p, MC_W >> !!p);
    }
    ic->rpl0[0] = ref[0]; ic->rpl0[1] = ref[1]; ic->rpl1[0] = ref[2]; ic->rpl1[1] = ref[0];
    static const uint8_t slot0[2] = { 0, 1 }, slot1[2] = { 2, 0 };
    for (int i = 0; i < 16; ++i) {
        ic->scale_fact_rpl0[i][0] = ic->scale_fact_rpl0[i][1] = 1 << RPR_SCALE_BITS;
        ic->scale_fact_rpl1[i][0] = ic->scale_fact_rpl1[i][1] = 1 << RPR_SCALE_BITS;
    }
    ic->prec_amvr = 0;                                        /* drv_affine_mvp.c:3508 */
    const struct OVBuffInfo *cb = &c->rcn_ctx.ctu_buff;
    struct shim_stream S;
    shim_stream_init(&S);
    if (g_shim) shim_bind(c, MC_W, MC_H, 0, 0);

    for (int l2w = 3; l2w <= 6; ++l2w) {
        for (int l2h = 3; l2h <= 6; ++l2h) {
            int w = 1 << l2w, h = 1 << l2h;
            if (w > MC_W || h > MC_H) continue;
            int reps = (w * h <= 256) ? 36 : (w * h <= 1024 ? 14 : 6);
            for (int rep = 0; rep < reps; ++rep) {
                ovhip_affine_desc d;
                memset(&d, 0, sizeof(d));
                int px, py;
                do {
                    px = rnd_range(0, (MC_W - w) / 4) * 4;
                    py = rnd_range(0, (MC_H - h) / 4) * 4;
                } while ((px >> 7) != ((px + w - 1) >> 7) || (py >> 7) != ((py + h - 1) >> 7));
                if (rep % 10 == 6) px = 0;
                if (rep % 10 == 8) py = MC_H - h;
                d.x0 = px; d.y0 = py; d.log2_w = l2w; d.log2_h = l2h;
                d.inter_dir = rnd_range(1, 3);
                int ri0 = rnd_range(0, 1), ri1 = rnd_range(0, 1);
                if (rep % 9 == 4) { d.inter_dir = 3; ri0 = 0; ri1 = 1; }          /* same picture in both lists */
                d.bcw_idx_plus1 = (rep % 4 == 1) ? rnd_range(1, 5) : 0;
                d.prof_dir = rep % 3 == 0 ? 0 : (d.inter_dir == 3 ? rnd_range(1, 3) : d.inter_dir);
                d.ref0 = slot0[ri0]; d.ref1 = slot1[ri1];
                d.poc0 = ic->rpl0[ri0]->poc; d.poc1 = ic->rpl1[ri1]->poc;
                d.mv_stride = w >> 2;
                for (int t = 0; t < 4; ++t)
                    for (int k = 0; k < 16; ++k) d.dmv_scale[t][k] = (int16_t)(rep % 5 == 3 ? (rnd_range(0, 1) ? 31 : -31) : rnd_range(-31, 31));

                /* a 6-parameter motion field per list, 1/16 pel; occasionally far outside the picture */
                int nsx = w >> 2, nsy = h >> 2;
                int32_t *mvs = calloc((size_t)nsx * nsy * 4, sizeof(int32_t));
                int32_t *m0 = mvs, *m1 = mvs + 2 * nsx * nsy;
                int range = rep % 8 == 0 ? 3000 : 300;
                for (int l = 0; l < 2; ++l) {
                    int bx = rnd_range(-range, range), by = rnd_range(-range, range);
                    int ax = rnd_range(-24, 24), ay = rnd_range(-24, 24), cx = rnd_range(-24, 24), cy = rnd_range(-24, 24);
                    if (rep % 7 == 2) { ax = ay = cx = cy = 0; bx &= ~15; by &= ~15; }          /* integer translation */
                    int32_t *m = l ? m1 : m0;
                    for (int j = 0; j < nsy; ++j)
                        for (int i = 0; i < nsx; ++i) {
                            m[2 * (j * nsx + i)] = bx + ((ax * i + cx * j) >> 1);
                            m[2 * (j * nsx + i) + 1] = by + ((ay * i + cy * j) >> 1);
                        }
                }
                if (rep % 9 == 4) memcpy(m1, m0, (size_t)nsx * nsy * 8);                       /* identical motion */
                d.mv0 = m0; d.mv1 = m1;

                c->ctb_x = px >> 7; c->ctb_y = py >> 7;
                int x0 = px & 127, y0 = py & 127;
                for (int j = 0; j < 128; ++j) memset(cb->y + j * cb->stride, 0xAB, 256);
                for (int j = 0; j < 64; ++j) { memset(cb->cb + j * cb->stride_c, 0xAB, 128); memset(cb->cr + j * cb->stride_c, 0xAB, 128); }

                struct PROFInfo pi;
                memcpy(&pi, d.dmv_scale, sizeof(pi));
                for (int j = 0; j < nsy; ++j)
                    for (int i = 0; i < nsx; ++i) {
                        int k = j * nsx + i;
                        OVMV mv0 = { .x = m0[2 * k], .y = m0[2 * k + 1], .ref_idx = ri0, .bcw_idx_plus1 = d.bcw_idx_plus1 };
                        OVMV mv1 = { .x = m1[2 * k], .y = m1[2 * k + 1], .ref_idx = ri1, .bcw_idx_plus1 = d.bcw_idx_plus1 };
                        if (!d.prof_dir)
                            c->rcn_funcs.rcn_mcp_b_l(c, *cb, ic, c->part_ctx, mv0, mv1, x0 + 4 * i, y0 + 4 * j, 2, 2, d.inter_dir, ri0, ri1);
                        else
                            c->rcn_funcs.rcn_prof_mcp_b_l(c, *cb, ic, c->part_ctx, mv0, mv1, x0 + 4 * i, y0 + 4 * j, 2, 2, d.inter_dir, ri0, ri1,
                                                          d.prof_dir, (const void *)&pi);
                    }
                for (int j = 0; j < nsy; j += 2)
                    for (int i = 0; i < nsx; i += 2) {
                        int k = j * nsx + i, k2 = k + nsx + 1;
                        OVMV mv0 = { .x = m0[2 * k] + m0[2 * k2], .y = m0[2 * k + 1] + m0[2 * k2 + 1], .ref_idx = ri0, .bcw_idx_plus1 = d.bcw_idx_plus1 };
                        OVMV mv1 = { .x = m1[2 * k] + m1[2 * k2], .y = m1[2 * k + 1] + m1[2 * k2 + 1], .ref_idx = ri1, .bcw_idx_plus1 = d.bcw_idx_plus1 };
                        mv0.x += mv0.x < 0; mv0.y += mv0.y < 0; mv0.x >>= 1; mv0.y >>= 1;
                        mv1.x += mv1.x < 0; mv1.y += mv1.y < 0; mv1.x >>= 1; mv1.y >>= 1;
                        c->rcn_funcs.rcn_mcp_b_c(c, *cb, ic, c->part_ctx, mv0, mv1, x0 + 4 * i, y0 + 4 * j, 3, 3, d.inter_dir, ri0, ri1);
                    }
                if (g_shim) shim_case_end(c, &S, "mca");

                uint32_t eoff[4];
                eoff[0] = (uint32_t)b_exp.n; dump_rect(&b_exp, cb->y, cb->stride, x0, y0, w, h);
                eoff[1] = (uint32_t)b_exp.n; dump_rect(&b_exp, cb->cb, cb->stride_c, x0 >> 1, y0 >> 1, w >> 1, h >> 1);
                eoff[2] = (uint32_t)b_exp.n; dump_rect(&b_exp, cb->cr, cb->stride_c, x0 >> 1, y0 >> 1, w >> 1, h >> 1);
                eoff[3] = (uint32_t)b_mv.n;
                gbuf_push(&b_mv, mvs, (size_t)nsx * nsy * 4);
                d.mv0 = d.mv1 = NULL;
                gbuf_push(&b_desc, &d, sizeof(d));
                gbuf_push(&b_eoff, eoff, 4);
                free(mvs);
                n_cases++;
            }
        }
    }

    if (g_shim) { shim_stream_write(dir, "shim_mca.ovg", &S, c, ref, 3); return; }
    gfile g = gfile_open(dir, "mca.ovg");
    uint32_t d3[3] = { 3, MC_H, MC_W };
    uint16_t *all = malloc(3 * MC_W * MC_H * 2);
    for (int i = 0; i < 3; ++i) memcpy(all + i * MC_W * MC_H, ref[i]->frame->data[0], MC_W * MC_H * 2);
    gfile_array(&g, "ref_y", T_U16, all, 3, d3);
    d3[1] = MC_H / 2; d3[2] = MC_W / 2;
    for (int i = 0; i < 3; ++i) memcpy(all + i * (MC_W / 2) * (MC_H / 2), ref[i]->frame->data[1], (MC_W / 2) * (MC_H / 2) * 2);
    gfile_array(&g, "ref_cb", T_U16, all, 3, d3);
    for (int i = 0; i < 3; ++i) memcpy(all + i * (MC_W / 2) * (MC_H / 2), ref[i]->frame->data[2], (MC_W / 2) * (MC_H / 2) * 2);
    gfile_array(&g, "ref_cr", T_U16, all, 3, d3);
    uint32_t d2[2] = { n_cases, sizeof(ovhip_affine_desc) };
    gfile_array(&g, "desc", T_U8, b_desc.data, 2, d2);
    d2[1] = 4; gfile_array(&g, "exp_off", T_U32, b_eoff.data, 2, d2);
    gfile_buf(&g, "exp", &b_exp);
    gfile_buf(&g, "mvs", &b_mv);
    gfile_close(&g);
    fprintf(stderr, "mca.ovg: %u cases, %zu expected samples\n", n_cases, b_exp.n);
}

/* ====================================================================================== LMCS
 * lmcs.ovg : rcn_init_lmcs (table construction), rcn_lmcs_compute_chroma_scale, lmcs_reshape_backward  -> K11
 * struct LMCSLUTs is private to rcn_lmcs.c:75-81. */
#include "rcn_lmcs.h"
struct LMCSLUTs { OVSample fwd_lut[1024]; OVSample bwd_lut[1024]; OVSample wnd_bnd[17]; };
#define LM_W 256
#define LM_H 128

static void
gen_lmcs(const char *dir)
{
    gbuf b_data = { .type = T_U8 }, b_luts = { .type = T_U8 }, b_reg = { .type = T_I32 }, b_inv = { .type = T_U16 };
    g_seed = 0x266 + 111;
    OVCTUDec *c = ref_new_ctudec(0, 1);
    const struct OVBuffInfo *cb = &c->rcn_ctx.ctu_buff;
    struct shim_stream S;
    shim_stream_init(&S);
    gbuf s_luts = { .type = T_U8 };
    if (g_shim) shim_bind(c, LM_W, LM_H, 0, 1);
    uint16_t *pic = malloc(LM_W * LM_H * 2);
    fill_plane(pic, LM_W, LM_H, LM_W);
    int n_sets = 0;

    for (int set = 0; set < 24; ++set) {
        struct OVLMCSData ld;
        memset(&ld, 0, sizeof(ld));
        ld.lmcs_min_bin_idx = rnd_range(0, 4);
        ld.lmcs_delta_max_bin_idx = rnd_range(0, 4);
        int amp = set % 4 == 0 ? 0 : (set % 4 == 1 ? 8 : (set % 4 == 2 ? 30 : 60));
        for (int i = 0; i < 16; ++i) {
            ld.lmcs_delta_abs_cw[i] = rnd_range(0, amp);
            ld.lmcs_delta_sign_cw_flag[i] = rnd_range(0, 1);
        }
        if (set % 6 == 5) { ld.lmcs_delta_abs_cw[7] = 64; ld.lmcs_delta_sign_cw_flag[7] = 1; }   /* an empty window */
        ld.lmcs_delta_abs_crs = rnd_range(0, 7);
        ld.lmcs_delta_sign_crs_flag = rnd_range(0, 1);

        struct LMCSInfo *li = &c->lmcs_info;
        if (!li->luts) li->luts = calloc(1, sizeof(struct LMCSLUTs));   /* keeps rcn_init_lmcs off ov_malloc (ovmem.c is not built) */
        c->rcn_funcs.rcn_init_lmcs(li, &ld);
        if (g_shim) gbuf_push(&s_luts, ovhip_shim_lmcs(c), sizeof(ovhip_lmcs_luts));

        ovhip_lmcs_data hd;
        memset(&hd, 0, sizeof(hd));
        hd.min_bin_idx = ld.lmcs_min_bin_idx; hd.delta_max_bin_idx = ld.lmcs_delta_max_bin_idx;
        hd.crs_offset = ld.lmcs_delta_sign_crs_flag ? -ld.lmcs_delta_abs_crs : ld.lmcs_delta_abs_crs;
        for (int i = 0; i < 16; ++i) hd.cw_delta[i] = ld.lmcs_delta_sign_cw_flag[i] ? -ld.lmcs_delta_abs_cw[i] : ld.lmcs_delta_abs_cw[i];
        gbuf_push(&b_data, &hd, sizeof(hd));

        ovhip_lmcs_luts hl;
        memset(&hl, 0, sizeof(hl));
        memcpy(hl.fwd_lut, li->luts->fwd_lut, 2048); memcpy(hl.bwd_lut, li->luts->bwd_lut, 2048);
        memcpy(hl.wnd_bnd, li->luts->wnd_bnd, 34);
        hl.min_idx = li->min_idx; hl.max_idx = li->max_idx; hl.crs_offset = li->lmcs_chroma_scaling_offset;
        gbuf_push(&b_luts, &hl, sizeof(hl));

        /* chroma scale of every 64x64 region of the 2-CTU picture under several availability patterns */
        for (int ctb = 0; ctb < 2; ++ctb) {
            /* CTU scratch = picture samples incl. the row above / column left of the CTU (zero outside the picture) */
            for (int j = -1; j < 128; ++j)
                for (int i = -1; i < 128; ++i) {
                    int px = ctb * 128 + i, py = j;
                    cb->y[j * cb->stride + i] = (px < 0 || py < 0) ? 0 : pic[py * LM_W + px];
                }
            for (int k = 0; k < 12; ++k) {
                int x0 = (k & 1) * 64, y0 = ((k >> 1) & 1) * 64;
                int n_abv = k < 4 ? 16 : rnd_range(0, 16), n_lft = k < 4 ? 16 : rnd_range(0, 16);
                if (y0 == 0) n_abv = 0;
                if (ctb == 0 && x0 == 0) n_lft = 0;
                struct CTUBitField pf;
                memset(&pf, 0, sizeof(pf));
                uint64_t am = n_abv ? ((1ull << n_abv) - 1) : 0, lm = n_lft ? ((1ull << n_lft) - 1) : 0;
                pf.hfield[y0 >> 2] = am << ((x0 >> 2) + 1);
                pf.vfield[x0 >> 2] = lm << ((y0 >> 2) + 1);
                li->lmcs_chroma_scale = 0;
                c->ctb_x = ctb; c->ctb_y = 0;
                c->rcn_funcs.rcn_lmcs_compute_chroma_scale(li, cb->stride, &pf, cb->y, x0, y0);
                if (g_shim) shim_case_end(c, &S, "lmcs region");
                int32_t rec[6] = { set, ctb * 128 + x0, y0, (int32_t)am, (int32_t)lm, li->lmcs_chroma_scale };
                gbuf_push(&b_reg, rec, 6);
            }
        }
        /* inverse mapping of the whole picture, CTU by CTU (every 6th table set keeps the fixture small) */
        for (int ctb = 0; ctb < 2 && set % 6 == 1; ++ctb) {
            uint16_t *blk = malloc(128 * 128 * 2);
            for (int j = 0; j < 128; ++j) memcpy(blk + j * 128, pic + j * LM_W + ctb * 128, 256);
            c->rcn_funcs.lmcs_reshape_backward(blk, 128, li->luts, 128, 128);
            gbuf_push(&b_inv, blk, 128 * 128);
            free(blk);
        }
        n_sets++;
    }

    if (g_shim) {
        gfile gs = gfile_open(dir, "shim_lmcs.ovg");
        uint32_t ds[2] = { (uint32_t)n_sets, sizeof(ovhip_lmcs_luts) };
        gfile_array(&gs, "luts", T_U8, s_luts.data, 2, ds);
        ds[0] = (uint32_t)(S.arr[OVHIP_REC_REGION].n / sizeof(ovhip_lmcs_region)); ds[1] = sizeof(ovhip_lmcs_region);
        gfile_array(&gs, "region", T_U8, S.arr[OVHIP_REC_REGION].data, 2, ds);
        gfile_close(&gs);
        fprintf(stderr, "shim_lmcs.ovg: %d table sets, %u regions\n", n_sets, ds[0]);
        return;
    }
    gfile g = gfile_open(dir, "lmcs.ovg");
    uint32_t d2[2] = { LM_H, LM_W };
    gfile_array(&g, "pic_y", T_U16, pic, 2, d2);
    d2[0] = n_sets; d2[1] = sizeof(ovhip_lmcs_data);  gfile_array(&g, "data", T_U8, b_data.data, 2, d2);
    d2[1] = sizeof(ovhip_lmcs_luts);                   gfile_array(&g, "luts", T_U8, b_luts.data, 2, d2);
    d2[0] = (uint32_t)(b_reg.n / 6); d2[1] = 6;        gfile_array(&g, "regions", T_I32, b_reg.data, 2, d2);
    uint32_t d4[4] = { (uint32_t)(b_inv.n / (2 * 128 * 128)), 2, 128, 128 };          gfile_array(&g, "inverse", T_U16, b_inv.data, 4, d4);
    gfile_close(&g);
    fprintf(stderr, "lmcs.ovg: %d table sets, %zu chroma-scale regions\n", n_sets, b_reg.n / 6);
}

/* ====================================================================================== GPM / CIIP
 * gpm.ovg : rcn_gpm_b (rcn_inter.c:3118-3143) for every merge_gpm_partition_idx, and rcn_ciip_b
 *           (rcn_inter.c:3011-3037) with the intra slots replaced by a stub that delivers a known
 *           "planar" prediction (intra prediction itself is outside the path)          -> K10 */
extern void rcn_init_gpm_params(void);
static uint16_t *g_ciip_intra[3];
static OVCTUDec *g_ciip_c;
static void stub_ciip_intra_l(const struct OVRCNCtx *const r, const struct OVBuffInfo *b, uint8_t m, int x0, int y0, int lw, int lh, CUFlags f)
{
    (void)r; (void)m; (void)f;
    int px = (g_ciip_c->ctb_x << 7) + x0, py = (g_ciip_c->ctb_y << 7) + y0;
    for (int j = 0; j < (1 << lh); ++j)
        for (int i = 0; i < (1 << lw); ++i) b->y[(y0 + j) * b->stride + x0 + i] = g_ciip_intra[0][(py + j) * MC_W + px + i];
}
static void stub_ciip_intra_c(const struct OVRCNCtx *const r, uint8_t m, int x0, int y0, int lw, int lh, CUFlags f)
{
    (void)m; (void)f;
    const struct OVBuffInfo *b = &r->ctu_buff;
    int px = (g_ciip_c->ctb_x << 6) + x0, py = (g_ciip_c->ctb_y << 6) + y0;
    for (int j = 0; j < (1 << lh); ++j)
        for (int i = 0; i < (1 << lw); ++i) {
            b->cb[(y0 + j) * b->stride_c + x0 + i] = g_ciip_intra[1][(py + j) * (MC_W / 2) + px + i];
            b->cr[(y0 + j) * b->stride_c + x0 + i] = g_ciip_intra[2][(py + j) * (MC_W / 2) + px + i];
        }
}

static void
gen_gpm(const char *dir)
{
    gbuf b_desc = { .type = T_U8 }, b_eoff = { .type = T_U32 }, b_exp = { .type = T_U16 }, b_ciip = { .type = T_I32 };
    uint32_t n_cases = 0, n_gpm = 0;
    g_seed = 0x266 + 131;
    rcn_init_gpm_params();

    OVCTUDec *c = ref_new_ctudec(0, 0);
    g_ciip_c = c;
    struct InterDRVCtx *ic = &c->drv_ctx.inter_ctx;
    OVPicture *ref[3];
    for (int i = 0; i < 3; ++i) {
        ref[i] = ref_new_picture(MC_W, MC_H, 8 * (i + 1));
        for (int p = 0; p < 3; ++p) fill_plane(ref[i]->frame->data[p], MC_W >> !!p, MC_H >> !!p, MC_W >> !!p);
    }
    for (int p = 0; p < 3; ++p) {
        g_ciip_intra[p] = malloc((MC_W >> !!p) * (MC_H >> !!p) * 2);
        fill_plane(g_ciip_intra[p], MC_W >> !!p, MC_H >> !!p, MC_W >> !!p);
    }
    ic->rpl0[0] = ref[0]; ic->rpl0[1] = ref[1]; ic->rpl1[0] = ref[2]; ic->rpl1[1] = ref[0];
    static const uint8_t slot[2][2] = { { 0, 1 }, { 2, 0 } };            /* [list][ref_idx] -> picture slot */
    for (int i = 0; i < 16; ++i) {
        ic->scale_fact_rpl0[i][0] = ic->scale_fact_rpl0[i][1] = 1 << RPR_SCALE_BITS;
        ic->scale_fact_rpl1[i][0] = ic->scale_fact_rpl1[i][1] = 1 << RPR_SCALE_BITS;
    }
    void *real_intra_l = (void *)c->rcn_funcs.intra_pred, *real_intra_c = (void *)c->rcn_funcs.intra_pred_c;
    c->rcn_funcs.intra_pred = stub_ciip_intra_l;
    c->rcn_funcs.intra_pred_c = stub_ciip_intra_c;
    uint16_t *cur[3];                                                     /* pass 2: the picture being decoded */
    for (int p = 0; p < 3; ++p) { cur[p] = malloc((MC_W >> !!p) * (MC_H >> !!p) * 2); }
    c->part_map.cu_mode_x = calloc(64, 1);
    const struct OVBuffInfo *cb = &c->rcn_ctx.ctu_buff;
    struct shim_stream S;
    shim_stream_init(&S);
    if (g_shim) shim_bind(c, MC_W, MC_H, 0, 0);

    uint32_t n_ciip2 = 0;
    for (int pass = 0; pass < 3; ++pass) {                                /* 0: GPM, 1: CIIP (planar stubbed), 2: CIIP, real planar */
        int n_iter = pass == 0 ? 64 * 3 : pass == 1 ? 150 : 120;
        if (pass == 2) {
            /* the real slots: intra_pred / intra_pred_c read the CTU scratch around the block, as in a decoder */
            c->rcn_funcs.intra_pred = real_intra_l; c->rcn_funcs.intra_pred_c = real_intra_c;
            c->rcn_funcs.rcn_attach_ctu_buff(&c->rcn_ctx, 7, 1);
            cb = &c->rcn_ctx.ctu_buff;
            for (int p = 0; p < 3; ++p) fill_plane(cur[p], MC_W >> !!p, MC_H >> !!p, MC_W >> !!p);
        }
        for (int it = 0; it < n_iter; ++it) {
            ovhip_pu_desc d;
            memset(&d, 0, sizeof(d));
            int l2w, l2h;
            if (pass == 0) { l2w = rnd_range(3, 6); l2h = rnd_range(3, 6); }
            else { do { l2w = rnd_range(2, 6); l2h = rnd_range(2, 6); } while (l2w + l2h < 6); }
            int w = 1 << l2w, h = 1 << l2h, px, py;
            do {
                px = rnd_range(0, (MC_W - w) / 4) * 4;
                py = rnd_range(0, (MC_H - h) / 4) * 4;
            } while ((px >> 7) != ((px + w - 1) >> 7) || (py >> 7) != ((py + h - 1) >> 7));
            d.x0 = px; d.y0 = py; d.log2_w = l2w; d.log2_h = l2h;
            int range = it % 9 == 0 ? 4000 : 400;
            d.mv0x = rnd_range(-range, range); d.mv0y = rnd_range(-range, range);
            d.mv1x = rnd_range(-range, range); d.mv1y = rnd_range(-range, range);
            if (it % 7 == 3) { d.mv0x &= ~15; d.mv1y &= ~15; }
            d.planes = 3;
            d.ref_idx0 = rnd_range(0, 1); d.ref_idx1 = rnd_range(0, 1);
            c->ctb_x = px >> 7; c->ctb_y = py >> 7;
            int x0 = px & 127, y0 = py & 127;
            if (pass < 2) {
                for (int j = 0; j < 128; ++j) memset(cb->y + j * cb->stride, 0xAB, 256);
                for (int j = 0; j < 64; ++j) { memset(cb->cb + j * cb->stride_c, 0xAB, 128); memset(cb->cr + j * cb->stride_c, 0xAB, 128); }
            } else {
                /* CTU scratch <- current picture (clamped at the picture border: what lies outside is never available);
                 * progress: the decoder's CTU start, every row of the CTU above the block, and the part of the block's rows
                 * (and some below) left of it */
                const int ox = c->ctb_x << 7, oy = c->ctb_y << 7;
                #define CLAMP(v, lo, hi) ((v) < (lo) ? (lo) : (v) > (hi) ? (hi) : (v))
                for (int j = -1; j < 132; ++j) for (int i = -128; i < 196; ++i)
                    cb->y[j * cb->stride + i] = cur[0][CLAMP(oy + j, 0, MC_H - 1) * MC_W + CLAMP(ox + i, 0, MC_W - 1)];
                for (int j = -1; j < 66; ++j) for (int i = -64; i < 98; ++i) {
                    const int yy = CLAMP(oy / 2 + j, 0, MC_H / 2 - 1), xx = CLAMP(ox / 2 + i, 0, MC_W / 2 - 1);
                    cb->cb[j * cb->stride_c + i] = cur[1][yy * (MC_W / 2) + xx]; cb->cr[j * cb->stride_c + i] = cur[2][yy * (MC_W / 2) + xx];
                }
                struct OVRCNCtx *r = &c->rcn_ctx;
                memset(&r->progress_field, 0, sizeof(r->progress_field)); memset(&r->progress_field_c, 0, sizeof(r->progress_field_c));
                init_ctu_bitfield(r, (c->ctb_x ? CTU_LFT_FLG : 0) | (c->ctb_y ? CTU_UP_FLG : 0), 7);
                const int xu = x0 >> 2, yu = y0 >> 2, hu = h >> 2;
                const int wu = (MC_W - ox) >> 2 < 32 ? (MC_W - ox) >> 2 : 32;          /* the CTU's columns inside the picture */
                if (yu) { ctu_field_set_rect_bitfield(&r->progress_field, 0, 0, wu, yu); ctu_field_set_rect_bitfield(&r->progress_field_c, 0, 0, wu, yu); }
                if (xu) {
                    int nl = rnd_range(hu, 2 * hu);
                    if (yu + nl > (MC_H >> 2)) nl = (MC_H >> 2) - yu;
                    if (yu + nl > 32) nl = 32 - yu;
                    ctu_field_set_rect_bitfield(&r->progress_field, 0, yu, xu, nl); ctu_field_set_rect_bitfield(&r->progress_field_c, 0, yu, xu, nl);
                }
            }
            int32_t cargs[2] = { 0, 0 };

            if (pass == 0) {
                struct VVCGPM *g = &ic->gpm_ctx;
                memset(g, 0, sizeof(*g));
                g->split_dir = it % 64;
                g->inter_dir0 = rnd_range(1, 2); g->inter_dir1 = rnd_range(1, 2);
                g->mv0 = (OVMV){ .x = d.mv0x, .y = d.mv0y, .ref_idx = d.ref_idx0 };
                g->mv1 = (OVMV){ .x = d.mv1x, .y = d.mv1y, .ref_idx = d.ref_idx1 };
                d.inter_dir = 3; d.refine = OVHIP_PU_GPM; d.gpm_split_dir = (uint8_t)g->split_dir;
                d.ref0 = slot[g->inter_dir0 - 1][d.ref_idx0]; d.ref1 = slot[g->inter_dir1 - 1][d.ref_idx1];
                d.prec_amvr_half = it % 5 == 2;
                ic->prec_amvr = d.prec_amvr_half ? MV_PRECISION_HALF : 0;
                c->rcn_funcs.rcn_gpm_b(c, g, x0, y0, l2w, l2h);
                n_gpm++;
            } else {
                d.inter_dir = rnd_range(1, 3);
                d.bcw_idx_plus1 = 0;
                d.poc0 = ic->rpl0[d.ref_idx0]->poc; d.poc1 = ic->rpl1[d.ref_idx1]->poc;
                d.ref0 = slot[0][d.ref_idx0]; d.ref1 = slot[1][d.ref_idx1];
                ic->prec_amvr = 0;
                cargs[0] = (int32_t[]){ OV_INTER, OV_INTRA, OV_MIP, OV_INTER_SKIP }[rnd_range(0, 3)];
                cargs[1] = (int32_t[]){ OV_INTER, OV_INTRA, OV_MIP, OV_INTER_SKIP }[rnd_range(0, 3)];
                c->part_map.cu_mode_x[(x0 + w - 1) >> 2] = (uint8_t)cargs[0];
                c->part_map.cu_mode_y[(y0 + h - 1) >> 2] = (uint8_t)cargs[1];
                OVMV mv0 = { .x = d.mv0x, .y = d.mv0y, .ref_idx = d.ref_idx0 };
                OVMV mv1 = { .x = d.mv1x, .y = d.mv1y, .ref_idx = d.ref_idx1 };
                if (d.inter_dir == 1 && it % 2)
                    c->rcn_funcs.rcn_ciip(c, x0, y0, l2w, l2h, mv0, d.ref_idx0);              /* P-slice entry point */
                else
                    c->rcn_funcs.rcn_ciip_b(c, mv0, mv1, x0, y0, l2w, l2h, d.inter_dir, d.ref_idx0, d.ref_idx1);
            }
            if (g_shim) shim_case_end(c, &S, "gpm / ciip");
            if (pass == 2) n_ciip2++;
            uint32_t eoff[3];
            eoff[0] = (uint32_t)b_exp.n; dump_rect(&b_exp, cb->y, cb->stride, x0, y0, w, h);
            eoff[1] = (uint32_t)b_exp.n; dump_rect(&b_exp, cb->cb, cb->stride_c, x0 >> 1, y0 >> 1, w >> 1, h >> 1);
            eoff[2] = (uint32_t)b_exp.n; dump_rect(&b_exp, cb->cr, cb->stride_c, x0 >> 1, y0 >> 1, w >> 1, h >> 1);
            gbuf_push(&b_desc, &d, sizeof(d));
            gbuf_push(&b_eoff, eoff, 3);
            gbuf_push(&b_ciip, cargs, 2);
            n_cases++;
        }
    }

    if (g_shim) { shim_stream_write(dir, "shim_gpm.ovg", &S, c, ref, 3); return; }
    gfile g = gfile_open(dir, "gpm.ovg");
    uint32_t d3[3] = { 3, MC_H, MC_W };
    uint16_t *all = malloc(3 * MC_W * MC_H * 2);
    for (int i = 0; i < 3; ++i) memcpy(all + i * MC_W * MC_H, ref[i]->frame->data[0], MC_W * MC_H * 2);
    gfile_array(&g, "ref_y", T_U16, all, 3, d3);
    d3[1] = MC_H / 2; d3[2] = MC_W / 2;
    for (int i = 0; i < 3; ++i) memcpy(all + i * (MC_W / 2) * (MC_H / 2), ref[i]->frame->data[1], (MC_W / 2) * (MC_H / 2) * 2);
    gfile_array(&g, "ref_cb", T_U16, all, 3, d3);
    for (int i = 0; i < 3; ++i) memcpy(all + i * (MC_W / 2) * (MC_H / 2), ref[i]->frame->data[2], (MC_W / 2) * (MC_H / 2) * 2);
    gfile_array(&g, "ref_cr", T_U16, all, 3, d3);
    uint32_t di[2] = { MC_H, MC_W };
    gfile_array(&g, "intra_y", T_U16, g_ciip_intra[0], 2, di);
    di[0] = MC_H / 2; di[1] = MC_W / 2;
    gfile_array(&g, "intra_cb", T_U16, g_ciip_intra[1], 2, di);
    gfile_array(&g, "intra_cr", T_U16, g_ciip_intra[2], 2, di);
    uint32_t d2[2] = { n_cases, sizeof(ovhip_pu_desc) };
    gfile_array(&g, "desc", T_U8, b_desc.data, 2, d2);
    d2[1] = 3; gfile_array(&g, "exp_off", T_U32, b_eoff.data, 2, d2);
    d2[1] = 2; gfile_array(&g, "ciip_modes", T_I32, b_ciip.data, 2, d2);
    uint32_t one = n_gpm; gfile_array(&g, "n_gpm", T_U32, &one, 1, (uint32_t[]){ 1 });
    one = n_ciip2; gfile_array(&g, "n_ciip_planar", T_U32, &one, 1, (uint32_t[]){ 1 });
    di[0] = MC_H; di[1] = MC_W; gfile_array(&g, "cur_y", T_U16, cur[0], 2, di);
    di[0] = MC_H / 2; di[1] = MC_W / 2; gfile_array(&g, "cur_cb", T_U16, cur[1], 2, di); gfile_array(&g, "cur_cr", T_U16, cur[2], 2, di);
    gfile_buf(&g, "exp", &b_exp);
    gfile_close(&g);
    fprintf(stderr, "gpm.ovg: %u GPM + %u CIIP (planar stubbed) + %u CIIP cases, %zu expected samples\n", n_gpm, n_cases - n_gpm - n_ciip2, n_ciip2, b_exp.n);
}

/* ====================================================================================== DBF */
#include "dbf_utils.h"
#include "drv_lines.h"
#include "slicedec.h"

static void
snapshot_dbf(ovhip_dbf_ctu *o, const struct DBFInfo *d)
{
    memset(o, 0, sizeof(*o));
    memcpy(o->ctb_bound_ver, d->ctb_bound_ver, sizeof(o->ctb_bound_ver));
    memcpy(o->ctb_bound_hor, d->ctb_bound_hor, sizeof(o->ctb_bound_hor));
    memcpy(o->ctb_bound_ver_c, d->ctb_bound_ver_c, sizeof(o->ctb_bound_ver_c));
    memcpy(o->ctb_bound_hor_c, d->ctb_bound_hor_c, sizeof(o->ctb_bound_hor_c));
    memcpy(o->aff_edg_ver, d->aff_edg_ver, sizeof(o->aff_edg_ver));
    memcpy(o->aff_edg_hor, d->aff_edg_hor, sizeof(o->aff_edg_hor));
    memcpy(o->bs2_ver, d->bs2_map.ver, sizeof(o->bs2_ver));       memcpy(o->bs2_hor, d->bs2_map.hor, sizeof(o->bs2_hor));
    memcpy(o->bs2c_ver, d->bs2_map_c.ver, sizeof(o->bs2c_ver));   memcpy(o->bs2c_hor, d->bs2_map_c.hor, sizeof(o->bs2c_hor));
    memcpy(o->bs1_ver, d->bs1_map.ver, sizeof(o->bs1_ver));       memcpy(o->bs1_hor, d->bs1_map.hor, sizeof(o->bs1_hor));
    memcpy(o->bs1cb_ver, d->bs1_map_cb.ver, sizeof(o->bs1cb_ver)); memcpy(o->bs1cb_hor, d->bs1_map_cb.hor, sizeof(o->bs1cb_hor));
    memcpy(o->bs1cr_ver, d->bs1_map_cr.ver, sizeof(o->bs1cr_ver)); memcpy(o->bs1cr_hor, d->bs1_map_cr.hor, sizeof(o->bs1cr_hor));
    memcpy(o->affine_ver, d->affine_map.ver, sizeof(o->affine_ver)); memcpy(o->affine_hor, d->affine_map.hor, sizeof(o->affine_hor));
    memcpy(o->qp_y, d->qp_map_y.hor, sizeof(o->qp_y));
    memcpy(o->qp_cb, d->qp_map_cb.hor, sizeof(o->qp_cb));
    memcpy(o->qp_cr, d->qp_map_cr.hor, sizeof(o->qp_cr));
    o->beta_offset = d->beta_offset; o->tc_offset = d->tc_offset;
    o->disable_v = d->disable_v; o->disable_h = d->disable_h;
}

/* random partition of a w x h region of one CTU into CUs; fills the deblocking maps the way the
 * decoder's parse loop does (vcl_coding_unit.c:742, vcl_transform_unit.c:1093-1146, :1934-1957,
 * rcn_transform_tree.c fill_bs_map calls, drv_affine_mvp.c:3052-3082) and paints a DC offset per CU
 * into the unfiltered picture so that block edges are visible to the filter decisions */
struct dbf_gen { struct DBFInfo *d; uint16_t *y, *cb, *cr; int stride, stride_c; int px, py; OVCTUDec *c; int bmode; struct IBCMVCtx *ibc; };

/* B-slice mode: the CU's motion goes into inter_ctx->mv_ctx0/1 (maps + 34x34 vectors, PB_POS_IN_BUF layout) the way
 * update_mv_ctx_b does; the MV-based bS pre-pass inside rcn_dbf_ctu then derives bS 1 itself */
static void
gen_cu_motion(struct dbf_gen *g, int x, int y, int w, int h, int intra)
{
    struct InterDRVCtx *ic = &g->c->drv_ctx.inter_ctx;
    int x0 = x >> 2, y0 = y >> 2, nw = w >> 2, nh = h >> 2;
    uint64_t mh = (((uint64_t)1 << nw) - 1) << (x0 + 1), mv_ = (((uint64_t)1 << nh) - 1) << (y0 + 1);
    if (intra) return;
    if (rnd_range(0, 15) == 0) {                       /* IBC CU: no inter motion, IBC map set */
        for (int j = 0; j < nh; ++j) g->ibc->ctu_map.hfield[y0 + 1 + j] |= mh;
        for (int i = 0; i < nw; ++i) g->ibc->ctu_map.vfield[x0 + 1 + i] |= mv_;
        return;
    }
    static const int pal[3][2] = { { 40, -24 }, { 44, -20 }, { -96, 130 } };
    int lists = rnd_range(1, 3);
    for (int l = 0; l < 2; ++l) {
        if (!(lists & (1 << l))) continue;
        struct OVMVCtx *m = l ? &ic->mv_ctx1 : &ic->mv_ctx0;
        int k = rnd_range(0, 2);
        OVMV mv = { .x = pal[k][0] + rnd_range(-6, 6), .y = pal[k][1] + rnd_range(-6, 6), .ref_idx = rnd_range(0, 2) };
        for (int j = 0; j < nh; ++j) { m->map.hfield[y0 + 1 + j] |= mh; for (int i = 0; i < nw; ++i) m->mvs[35 + x0 + i + (y0 + j) * 34] = mv; }
        for (int i = 0; i < nw; ++i) m->map.vfield[x0 + 1 + i] |= mv_;
    }
}

static void
gen_cu(struct dbf_gen *g, int x, int y, int w, int h)
{
    struct DBFInfo *d = g->d;
    int l2w = 31 - __builtin_clz(w), l2h = 31 - __builtin_clz(h);
    int qp = rnd_range(18, 50);
    int intra = rnd_range(0, 9) == 0;
    int affine = !intra && w >= 16 && h >= 16 && rnd_range(0, 5) == 0;
    int off_y = rnd_range(-40, 40), off_c = rnd_range(-24, 24);
    for (int j = 0; j < h; ++j) for (int i = 0; i < w; ++i) {
        int v = g->y[(g->py + y + j) * g->stride + g->px + x + i] + off_y;
        g->y[(g->py + y + j) * g->stride + g->px + x + i] = v < 0 ? 0 : v > 1023 ? 1023 : v;
    }
    for (int j = 0; j < h / 2; ++j) for (int i = 0; i < w / 2; ++i) {
        int o = ((g->py + y) / 2 + j) * g->stride_c + (g->px + x) / 2 + i;
        int v = g->cb[o] + off_c; g->cb[o] = v < 0 ? 0 : v > 1023 ? 1023 : v;
        v = g->cr[o] - off_c;     g->cr[o] = v < 0 ? 0 : v > 1023 ? 1023 : v;
    }
    dbf_fill_cu_edge(&d->cu_edge, x >> 2, y >> 2, w >> 2, h >> 2);
    if (intra) { fill_bs_map(&d->bs2_map, x, y, l2w, l2h); fill_bs_map(&d->bs2_map_c, x, y, l2w, l2h); }
    if (g->bmode) gen_cu_motion(g, x, y, w, h, intra);
    else if (rnd_range(0, 2) == 0) {      /* "motion differs from the neighbours" (what dbf_ctu_preproc_* derives) */
        fill_bs_map(&d->bs1_map, x, y, l2w, l2h);
        if (rnd_range(0, 1)) fill_bs_map(&d->bs1_map_cb, x, y, l2w, l2h);
        if (rnd_range(0, 1)) fill_bs_map(&d->bs1_map_cr, x, y, l2w, l2h);
    }
    if (affine) {
        int x0_u = x >> 2, y0_u = y >> 2, nw = w >> 2, nh = h >> 2;
        uint64_t msk_v = (((uint64_t)1 << nh) - 1) << y0_u, msk_h = (((uint64_t)1 << nw) - 1) << (2 + x0_u);
        for (int i = 2; i < nw; i += 2) { d->aff_edg_ver[8 + x0_u + i] |= msk_v; if (rnd_range(0, 1)) d->bs1_map.ver[x0_u + i] |= msk_v & ((uint64_t)rnd32() << 32 | rnd32() << 8 | rnd32()); }
        for (int i = 2; i < nh; i += 2) { d->aff_edg_hor[8 + y0_u + i] |= msk_h; if (rnd_range(0, 1)) d->bs1_map.hor[y0_u + i] |= msk_h & ((uint64_t)rnd32() << 32 | rnd32() << 8 | rnd32()); }
        dbf_fill_aff_map(&d->affine_map, x0_u, y0_u, nw, nh);
    }
    /* transform units: split at 64 */
    for (int ty = 0; ty < h; ty += 64) for (int tx = 0; tx < w; tx += 64) {
        int tl2w = l2w > 6 ? 6 : l2w, tl2h = l2h > 6 ? 6 : l2h;
        fill_ctb_bound(d, x + tx, y + ty, tl2w, tl2h);
        fill_ctb_bound_c(d, x + tx, y + ty, tl2w, tl2h);
        dbf_fill_qp_map(&d->qp_map_y, x + tx, y + ty, tl2w, tl2h, qp);
        dbf_fill_qp_map(&d->qp_map_cb, x + tx, y + ty, tl2w, tl2h, qp - rnd_range(0, 3));
        dbf_fill_qp_map(&d->qp_map_cr, x + tx, y + ty, tl2w, tl2h, qp - rnd_range(0, 3));
        if (rnd_range(0, 1)) fill_bs_map(&d->bs1_map, x + tx, y + ty, tl2w, tl2h);          /* cbf luma */
        if (rnd_range(0, 2) == 0) fill_bs_map(&d->bs1_map_cb, x + tx, y + ty, tl2w, tl2h);  /* cbf cb   */
        if (rnd_range(0, 2) == 0) fill_bs_map(&d->bs1_map_cr, x + tx, y + ty, tl2w, tl2h);  /* cbf cr   */
    }
}

static void
gen_part(struct dbf_gen *g, int x, int y, int w, int h, int lim_w, int lim_h)
{
    if (x >= lim_w || y >= lim_h) return;
    if (x + w > lim_w || y + h > lim_h) {
        if (w > 4 && x + w > lim_w && !(h > 4 && y + h > lim_h)) { gen_part(g, x, y, w / 2, h, lim_w, lim_h); gen_part(g, x + w / 2, y, w / 2, h, lim_w, lim_h); }
        else if (h > 4 && y + h > lim_h && !(w > 4 && x + w > lim_w)) { gen_part(g, x, y, w, h / 2, lim_w, lim_h); gen_part(g, x, y + h / 2, w, h / 2, lim_w, lim_h); }
        else { gen_part(g, x, y, w / 2, h / 2, lim_w, lim_h); gen_part(g, x + w / 2, y, w / 2, h / 2, lim_w, lim_h);
               gen_part(g, x, y + h / 2, w / 2, h / 2, lim_w, lim_h); gen_part(g, x + w / 2, y + h / 2, w / 2, h / 2, lim_w, lim_h); }
        return;
    }
    int m = w > h ? w : h;
    static const int p_split[8] = { 0, 0, 0, 15, 45, 65, 85, 95 };   /* % by log2(max dim): 8->15 .. 128->95 */
    if (m > 4 && rnd_range(0, 99) < p_split[31 - __builtin_clz(m)]) {
        int k = rnd_range(0, 3);
        if (k < 2 && w == h && w > 4) { gen_part(g, x, y, w / 2, h / 2, lim_w, lim_h); gen_part(g, x + w / 2, y, w / 2, h / 2, lim_w, lim_h);
                                        gen_part(g, x, y + h / 2, w / 2, h / 2, lim_w, lim_h); gen_part(g, x + w / 2, y + h / 2, w / 2, h / 2, lim_w, lim_h); }
        else if ((k == 2 && w > 4) || h <= 4) { gen_part(g, x, y, w / 2, h, lim_w, lim_h); gen_part(g, x + w / 2, y, w / 2, h, lim_w, lim_h); }
        else { gen_part(g, x, y, w, h / 2, lim_w, lim_h); gen_part(g, x, y + h / 2, w, h / 2, lim_w, lim_h); }
        return;
    }
    gen_cu(g, x, y, w, h);
}

static void
gen_dbf(const char *dir)
{
    enum { NPIC = 3 };
    static const int PW[NPIC] = { 304, 264, 256 }, PH[NPIC] = { 200, 136, 192 };     /* picture 2: B slice, motion-derived bS */
    gfile g = gfile_open(dir, g_shim ? "shim_dbf.ovg" : "dbf.ovg");
    g_seed = 0x266 + 2;
    for (int pi = 0; pi < NPIC; ++pi) {
        const int W = PW[pi], H = PH[pi], nx = (W + 127) / 128, ny = (H + 127) / 128;
        uint16_t *y = malloc(W * H * 2), *cb = malloc(W * H / 2), *cr = malloc(W * H / 2);
        fill_plane(y, W, H, W); fill_plane(cb, W / 2, H / 2, W / 2); fill_plane(cr, W / 2, H / 2, W / 2);
        /* very smooth content so that beta decisions pass often: low-pass once more */
        for (int j = 0; j < H; ++j) for (int i = 1; i < W; ++i) y[j * W + i] = (y[j * W + i] + 3 * y[j * W + i - 1] + 2) >> 2;
        for (int j = 1; j < H; ++j) for (int i = 0; i < W; ++i) y[j * W + i] = (y[j * W + i] + 3 * y[(j - 1) * W + i] + 2) >> 2;
        for (int j = 0; j < H / 2; ++j) for (int i = 1; i < W / 2; ++i) { cb[j * (W / 2) + i] = (cb[j * (W / 2) + i] + 3 * cb[j * (W / 2) + i - 1] + 2) >> 2; cr[j * (W / 2) + i] = (cr[j * (W / 2) + i] + 3 * cr[j * (W / 2) + i - 1] + 2) >> 2; }
        for (int j = 1; j < H / 2; ++j) for (int i = 0; i < W / 2; ++i) { cb[j * (W / 2) + i] = (cb[j * (W / 2) + i] + 3 * cb[(j - 1) * (W / 2) + i] + 2) >> 2; cr[j * (W / 2) + i] = (cr[j * (W / 2) + i] + 3 * cr[(j - 1) * (W / 2) + i] + 2) >> 2; }

        OVCTUDec *c = ref_new_ctudec(0, 0);
        if (g_shim) shim_bind(c, W, H, 0, 0);
        struct DBFInfo *d = &c->dbf_info;
        const int bmode = pi == 2;
        c->tmp_slice_type = bmode ? 0 : 2;     /* I: the MV-based bS pre-pass is emulated by gen_cu(); B: rcn_dbf_ctu derives it */
        static struct IBCMVCtx ibc;
        d->ibc_ctx = &ibc;
        struct InterDRVCtx *ic = &c->drv_ctx.inter_ctx;
        static const int16_t dr0[16] = { -8, -16, 8, -24 }, dr1[16] = { 8, -8, 16, -16 };   /* POC distances: lists share pictures */
        memcpy(ic->dist_ref_0, dr0, sizeof(dr0)); memcpy(ic->dist_ref_1, dr1, sizeof(dr1));
        gbuf b_mv = { .type = T_U8 };
        struct DBFLines L;
        int npu = W / 4 + 40;
        L.qp_x_map = calloc(npu, 1); L.qp_x_map_cb = calloc(npu, 1); L.qp_x_map_cr = calloc(npu, 1);
        L.small_map = calloc(nx + 1, 8); L.dbf_bs2_hor = calloc(nx + 1, 8); L.dbf_bs2_hor_c = calloc(nx + 1, 8);
        L.dbf_bs1_hor = calloc(nx + 1, 8); L.dbf_bs1_hor_cb = calloc(nx + 1, 8); L.dbf_bs1_hor_cr = calloc(nx + 1, 8);
        L.dbf_affine = calloc(nx + 1, 8); L.large_map_c = calloc(nx + 1, 8);

        /* the partition pass paints block offsets into the picture, so do it first for all CTUs
         * into per-CTU map snapshots?  No: maps are CTU-local and rotated by load/store, so generate,
         * filter and snapshot CTU by CTU exactly in decoding order; the unfiltered picture is captured
         * from a second identical generation pass (same seed) that does not filter. */
        uint16_t *y0 = malloc(W * H * 2), *cb0 = malloc(W * H / 2), *cr0 = malloc(W * H / 2);
        gbuf b_ctu = { .type = T_U8 };
        uint32_t n_ctu = 0;
        for (int pass = 0; pass < 2; ++pass) {
            uint32_t seed_save = g_seed;
            uint16_t *ty = pass ? y : y0, *tcb = pass ? cb : cb0, *tcr = pass ? cr : cr0;
            if (!pass) { memcpy(y0, y, W * H * 2); memcpy(cb0, cb, W * H / 2); memcpy(cr0, cr, W * H / 2); }
            memset(d, 0, sizeof(*d));
            d->ibc_ctx = &ibc;
            d->beta_offset = (pi ? 2 : -2) * 2; d->tc_offset = (pi ? -1 : 3) * 2;
            memset(L.qp_x_map, 0, npu); memset(L.qp_x_map_cb, 0, npu); memset(L.qp_x_map_cr, 0, npu);
            memset(L.small_map, 0, (nx + 1) * 8); memset(L.dbf_bs2_hor, 0, (nx + 1) * 8); memset(L.dbf_bs2_hor_c, 0, (nx + 1) * 8);
            memset(L.dbf_bs1_hor, 0, (nx + 1) * 8); memset(L.dbf_bs1_hor_cb, 0, (nx + 1) * 8); memset(L.dbf_bs1_hor_cr, 0, (nx + 1) * 8);
            memset(L.dbf_affine, 0, (nx + 1) * 8); memset(L.large_map_c, 0, (nx + 1) * 8);
            for (int cy = 0; cy < ny; ++cy) {
                dbf_load_info(d, &L, 7, 0);
                for (int cx = 0; cx < nx; ++cx) {
                    int ctu_w = W - cx * 128 < 128 ? W - cx * 128 : 128, ctu_h = H - cy * 128 < 128 ? H - cy * 128 : 128;
                    struct dbf_gen gg = { d, ty, tcb, tcr, W, W / 2, cx * 128, cy * 128, c, bmode, &ibc };
                    if (bmode) {
                        /* fresh CTU motion context; the 1-unit border (row above / column left) stands in for what
                         * the line buffers would bring in from the neighbouring CTUs */
                        memset(&ic->mv_ctx0, 0, sizeof(ic->mv_ctx0)); memset(&ic->mv_ctx1, 0, sizeof(ic->mv_ctx1)); memset(&ibc, 0, sizeof(ibc));
                        for (int l = 0; l < 2; ++l) {
                            struct OVMVCtx *m = l ? &ic->mv_ctx1 : &ic->mv_ctx0;
                            for (int k = 0; k < 33; ++k) {
                                if (cy && rnd_range(0, 2)) { m->map.hfield[0] |= (uint64_t)1 << (k + 1); m->map.vfield[k + 1] |= 1;
                                                             m->mvs[1 + k] = (OVMV){ .x = 40 + rnd_range(-8, 8), .y = -24 + rnd_range(-8, 8), .ref_idx = rnd_range(0, 2) }; }
                                if (cx && rnd_range(0, 2)) { m->map.vfield[0] |= (uint64_t)1 << (k + 1); m->map.hfield[k + 1] |= 1;
                                                             m->mvs[34 + 34 * k] = (OVMV){ .x = 44 + rnd_range(-8, 8), .y = -20 + rnd_range(-8, 8), .ref_idx = rnd_range(0, 2) }; }
                            }
                        }
                    }
                    gen_part(&gg, 0, 0, 128, 128, ctu_w, ctu_h);
                    if (pass) {
                        c->ctu_ngh_flags = (cx ? CTU_LFT_FLG : 0) | (cy ? CTU_UP_FLG : 0);
                        c->ctb_x = cx; c->ctb_y = cy;
                        c->rcn_ctx.frame_buff.y = y + cy * 128 * W + cx * 128;
                        c->rcn_ctx.frame_buff.cb = cb + cy * 64 * (W / 2) + cx * 64;
                        c->rcn_ctx.frame_buff.cr = cr + cy * 64 * (W / 2) + cx * 64;
                        c->rcn_ctx.frame_buff.stride = W; c->rcn_ctx.frame_buff.stride_c = W / 2;
                        int last_x = cx == nx - 1, last_y = cy == ny - 1;
                        int truncated = ctu_w < 128 || ctu_h < 128;
                        ovhip_dbf_ctu o;
                        snapshot_dbf(&o, d);            /* BEFORE the slot: in a B slice it adds the motion-derived bS 1 to d */
                        if (bmode) {
                            ovhip_dbf_mv_ctx mc;
                            memset(&mc, 0, sizeof(mc));
                            memcpy(mc.cu_edge_ver, d->cu_edge.ver, sizeof(mc.cu_edge_ver)); memcpy(mc.cu_edge_hor, d->cu_edge.hor, sizeof(mc.cu_edge_hor));
                            memcpy(mc.map0_h, ic->mv_ctx0.map.hfield, sizeof(mc.map0_h)); memcpy(mc.map0_v, ic->mv_ctx0.map.vfield, sizeof(mc.map0_v));
                            memcpy(mc.map1_h, ic->mv_ctx1.map.hfield, sizeof(mc.map1_h)); memcpy(mc.map1_v, ic->mv_ctx1.map.vfield, sizeof(mc.map1_v));
                            memcpy(mc.ibc_h, ibc.ctu_map.hfield, sizeof(mc.ibc_h)); memcpy(mc.ibc_v, ibc.ctu_map.vfield, sizeof(mc.ibc_v));
                            memcpy(mc.dist_ref0, ic->dist_ref_0, sizeof(mc.dist_ref0)); memcpy(mc.dist_ref1, ic->dist_ref_1, sizeof(mc.dist_ref1));
                            mc.mv_bytes = sizeof(OVMV);
                            gbuf_push(&b_mv, &mc, sizeof(mc));
                            gbuf_push(&b_mv, ic->mv_ctx0.mvs, sizeof(ic->mv_ctx0.mvs));
                            gbuf_push(&b_mv, ic->mv_ctx1.mvs, sizeof(ic->mv_ctx1.mvs));
                        }
                        if (!truncated) TIMED(TS_DBF, c->rcn_funcs.df.rcn_dbf_ctu(&c->rcn_ctx, d, 7, last_x, last_y));
                        else            TIMED(TS_DBF, c->rcn_funcs.df.rcn_dbf_truncated_ctu(&c->rcn_ctx, d, 7, last_x, last_y, ctu_w, ctu_h));
                        o.log2_ctu_s = 7; o.last_x = last_x; o.last_y = last_y;
                        o.ctu_lft = !!cx; o.ctu_abv = !!cy;
                        o.ctu_w = truncated ? ctu_w : 0; o.ctu_h = truncated ? ctu_h : 0;
                        o.ctb_x = cx; o.ctb_y = cy;
                        gbuf_push(&b_ctu, &o, sizeof(o));
                        n_ctu++;
                    }
                    dbf_store_info(d, &L, 7, cx);
                    if (cx < nx - 1) dbf_load_info(d, &L, 7, cx + 1);
                }
            }
            if (!pass) g_seed = seed_save;      /* replay the same partition in the filtering pass */
        }
        /* pass 0 painted y0; pass 1 painted y identically then filtered it */
        char nm[32];
        if (g_shim) {
            /* what the installed df.rcn_dbf_ctu / rcn_dbf_truncated_ctu slots recorded for the whole picture */
            if (ovhip_shim_last_error(c)) { fprintf(stderr, "shim: dbf picture %d latched %d\n", pi, ovhip_shim_last_error(c)); exit(1); }
            ovhip_recorder *r = ovhip_shim_recorder(c);
            ovhip_dbf_offsets offs;
            for (int dir2 = 0; dir2 < 2; ++dir2) {
                size_t ne = 0;
                const ovhip_dbf_edge *ed = ovhip_rec_dbf_edges(r, dir2, &ne, &offs);
                uint32_t de[2] = { (uint32_t)ne, sizeof(ovhip_dbf_edge) };
                snprintf(nm, 32, "p%d_edges_%c", pi, dir2 ? 'h' : 'v'); gfile_array(&g, nm, T_U8, ne ? (const void *)ed : (const void *)"", 2, de);
            }
            uint32_t dof = sizeof(offs);
            snprintf(nm, 32, "p%d_offsets", pi); gfile_array(&g, nm, T_I8, &offs, 1, &dof);
            continue;
        }
        uint32_t d2[2] = { H, W };
        snprintf(nm, 32, "p%d_in_y", pi); gfile_array(&g, nm, T_U16, y0, 2, d2);
        snprintf(nm, 32, "p%d_exp_y", pi); gfile_array(&g, nm, T_U16, y, 2, d2);
        d2[0] = H / 2; d2[1] = W / 2;
        snprintf(nm, 32, "p%d_in_cb", pi); gfile_array(&g, nm, T_U16, cb0, 2, d2);
        snprintf(nm, 32, "p%d_in_cr", pi); gfile_array(&g, nm, T_U16, cr0, 2, d2);
        snprintf(nm, 32, "p%d_exp_cb", pi); gfile_array(&g, nm, T_U16, cb, 2, d2);
        snprintf(nm, 32, "p%d_exp_cr", pi); gfile_array(&g, nm, T_U16, cr, 2, d2);
        d2[0] = n_ctu; d2[1] = sizeof(ovhip_dbf_ctu);
        snprintf(nm, 32, "p%d_ctus", pi); gfile_array(&g, nm, T_U8, b_ctu.data, 2, d2);
        if (bmode) { d2[1] = (uint32_t)(b_mv.n / n_ctu); snprintf(nm, 32, "p%d_mvctx", pi); gfile_array(&g, nm, T_U8, b_mv.data, 2, d2); }
        size_t diff = 0;
        for (int i = 0; i < W * H; ++i) diff += y[i] != y0[i];
        fprintf(stderr, "dbf.ovg: picture %d %dx%d, %u CTUs, %zu luma samples changed by the reference\n", pi, W, H, n_ctu, diff);
    }
    gfile_close(&g);
}


/* ====================================================================================== SAO */
/* rcn_alloc_filter_buffers() (rcn_ctu.c:512-551) calls ov_malloc (ovmem.c is not built, see
 * oracle/Makefile); the harness performs the same sizing with plain calloc. */
static void
harness_alloc_filter_buffers(struct OVRCNCtx *rcn_ctx, int nb_ctu_w, int margin, int log2_ctb_s)
{
    struct OVFilterBuffers *fb = &rcn_ctx->filter_buffers;
    int ctu_s = 1 << log2_ctb_s;
    fb->margin = margin;
    for (int comp = 0; comp < 3; ++comp) {
        int ratio = comp ? 2 : 1;
        fb->filter_region_w[comp] = ctu_s / ratio;
        fb->filter_region_h[comp] = ctu_s / ratio;
        fb->filter_region_stride[comp] = ctu_s / ratio + 2 * margin;
        fb->filter_region_offset[comp] = margin * fb->filter_region_stride[comp] + margin;
        int ext = fb->filter_region_stride[comp] * (fb->filter_region_h[comp] + 2 * margin + 1);
        fb->filter_region[comp] = calloc(ext, sizeof(OVSample));
        fb->saved_cols[comp] = calloc(fb->filter_region_h[comp] * margin, sizeof(OVSample));
        fb->saved_rows_stride[comp] = nb_ctu_w * ctu_s / ratio;
        fb->saved_rows_sao[comp] = calloc(margin * fb->saved_rows_stride[comp], sizeof(OVSample));
        fb->saved_rows_alf[comp] = calloc(margin * fb->saved_rows_stride[comp], sizeof(OVSample));
    }
}

static OVFrame *
harness_frame(int w, int h)
{
    OVPicture *p = ref_new_picture(w, h, 0);
    fill_plane(p->frame->data[0], w, h, w);
    fill_plane(p->frame->data[1], w / 2, h / 2, w / 2);
    fill_plane(p->frame->data[2], w / 2, h / 2, w / 2);
    return p->frame;
}

static void
gen_sao(const char *dir)
{
    enum { NPIC = 3 };
    static const int PW[NPIC] = { 304, 264, 136 }, PH[NPIC] = { 200, 264, 72 };
    gfile g = gfile_open(dir, g_shim ? "shim_sao.ovg" : "sao.ovg");
    g_seed = 0x266 + 3;
    for (int pi = 0; pi < NPIC; ++pi) {
        const int W = PW[pi], H = PH[pi], nx = (W + 127) / 128, ny = (H + 127) / 128;
        OVFrame *f = harness_frame(W, H);
        /* sprinkle extremes so clipping and every band are hit */
        uint16_t *py = f->data[0];
        for (int i = 0; i < W * H; i += 11) py[i] = (uint16_t)rnd_range(0, 1023);
        char nm[32];
        uint32_t d2[2] = { H, W };
        if (!g_shim) {
        snprintf(nm, 32, "p%d_in_y", pi); gfile_array(&g, nm, T_U16, f->data[0], 2, d2);
        d2[0] = H / 2; d2[1] = W / 2;
        snprintf(nm, 32, "p%d_in_cb", pi); gfile_array(&g, nm, T_U16, f->data[1], 2, d2);
        snprintf(nm, 32, "p%d_in_cr", pi); gfile_array(&g, nm, T_U16, f->data[2], 2, d2);
        }

        OVCTUDec *c = ref_new_ctudec(0, 0);
        if (g_shim) shim_bind(c, W, H, 0, 0);
        c->pic_w = W; c->pic_h = H;
        c->rcn_ctx.frame_start = f;
        harness_alloc_filter_buffers(&c->rcn_ctx, nx, 3, 7);
        c->sao_info.sao_luma_flag = 1; c->sao_info.sao_chroma_flag = 1; c->sao_info.chroma_format_idc = 1;
        SAOParamsCtu *prm = calloc(nx * ny, sizeof(*prm));
        ovhip_sao_ctu *mine = calloc(nx * ny, sizeof(*mine));
        c->sao_info.sao_params = prm;
        for (int i = 0; i < nx * ny; ++i) {
            for (int comp = 0; comp < 3; ++comp) {
                int t = rnd_range(0, 3); if (t == 3) t = 2;
                prm[i].type_idx[comp] = t;
                prm[i].band_position[comp] = rnd_range(0, 31);
                prm[i].eo_class[comp] = rnd_range(0, 3);
                for (int k = 0; k < 5; ++k) prm[i].offset_val[comp][k] = (int16_t)rnd_range(-31, 31);
                if (t == 2) { prm[i].offset_val[comp][2] = 0; }
                mine[i].type[comp] = t; mine[i].band_position[comp] = prm[i].band_position[comp];
                mine[i].eo_class[comp] = prm[i].eo_class[comp];
                memcpy(mine[i].offset_val[comp], prm[i].offset_val[comp], 10);
            }
        }
        struct RectEntryInfo einfo;
        memset(&einfo, 0, sizeof(einfo));
        einfo.nb_ctu_w = nx; einfo.nb_ctu_h = ny;
        /* call order of decode_ctu_line / decode_ctu_last_line (slicedec.c:934-956, :1058-1073) */
        for (int cy = 0; cy < ny; ++cy) {
            c->ctb_y = cy;
            if (cy == 0) {
                TIMED(TS_SAO, c->rcn_funcs.sao.rcn_sao_first_pix_rows(c, &einfo, 0));
                if (ny == 1) TIMED(TS_SAO, c->rcn_funcs.sao.rcn_sao_filter_line(c, &einfo, 0));
            } else if (cy == ny - 1) {
                TIMED(TS_SAO, c->rcn_funcs.sao.rcn_sao_filter_line(c, &einfo, cy - 1));
                TIMED(TS_SAO, c->rcn_funcs.sao.rcn_sao_filter_line(c, &einfo, cy));
            } else {
                TIMED(TS_SAO, c->rcn_funcs.sao.rcn_sao_filter_line(c, &einfo, cy - 1));
            }
        }
        if (g_shim) {
            size_t nc = 0;
            const ovhip_sao_ctu *sp = ovhip_shim_sao_params(c, &nc);
            if (!sp || nc != (size_t)(nx * ny) || ovhip_shim_last_error(c)) { fprintf(stderr, "shim: sao picture %d: no parameters captured\n", pi); exit(1); }
            d2[0] = nx * ny; d2[1] = sizeof(ovhip_sao_ctu);
            snprintf(nm, 32, "p%d_params", pi); gfile_array(&g, nm, T_U8, sp, 2, d2);
            continue;
        }
        d2[0] = H; d2[1] = W;
        snprintf(nm, 32, "p%d_exp_y", pi); gfile_array(&g, nm, T_U16, f->data[0], 2, d2);
        d2[0] = H / 2; d2[1] = W / 2;
        snprintf(nm, 32, "p%d_exp_cb", pi); gfile_array(&g, nm, T_U16, f->data[1], 2, d2);
        snprintf(nm, 32, "p%d_exp_cr", pi); gfile_array(&g, nm, T_U16, f->data[2], 2, d2);
        d2[0] = nx * ny; d2[1] = sizeof(ovhip_sao_ctu);
        snprintf(nm, 32, "p%d_params", pi); gfile_array(&g, nm, T_U8, mine, 2, d2);
        fprintf(stderr, "sao.ovg: picture %d %dx%d (%d CTUs)\n", pi, W, H, nx * ny);
    }
    if (g_shim) {
        /* a picture cut into TWO rect entries (tile columns) with each entry on its OWN OVCTUDec (entry threads: ovthreads.c:112-114),
         * record-only: the shim keeps one recorder per OVCTUDec.  The entry that starts the picture is taken; the one whose
         * OVCTUDec never saw the picture's first entry has nobody to record into and must latch OVHIP_EUNSUP at its attach (the
         * entries of a picture in turn on ONE OVCTUDec are the supported form: gen_pipe's tiles streams). */
        const int W = 384, H = 136, nx = 3, ny = 2;
        int32_t codes[2];
        for (int en = 0; en < 2; ++en) {
            OVFrame *f = harness_frame(W, H);
            OVCTUDec *c = ref_new_ctudec(0, 0);
            shim_bind(c, W, H, 0, 0);
            c->pic_w = W; c->pic_h = H;
            c->rcn_ctx.frame_start = f;
            harness_alloc_filter_buffers(&c->rcn_ctx, nx, 3, 7);
            c->sao_info.sao_luma_flag = 1; c->sao_info.chroma_format_idc = 1;
            c->sao_info.sao_params = calloc(nx * ny, sizeof(SAOParamsCtu));
            struct RectEntryInfo einfo;
            memset(&einfo, 0, sizeof(einfo));
            einfo.ctb_x = en ? 2 : 0; einfo.nb_ctu_w = en ? 1 : 2; einfo.nb_ctu_h = ny;
            c->ctb_y = 0;
            c->rcn_funcs.rcn_attach_frame_buff(&c->rcn_ctx, f, &einfo, 7);
            c->rcn_funcs.sao.rcn_sao_first_pix_rows(c, &einfo, 0);
            codes[en] = ovhip_shim_last_error(c);
            if (codes[en] != (en ? OVHIP_EUNSUP : 0)) { fprintf(stderr, "shim: entry %d of a two-OVCTUDec picture: code %d\n", en, codes[en]); exit(1); }
        }
        uint32_t d1 = 2;
        gfile_array(&g, "two_entries_latched", T_I32, codes, 1, &d1);
        fprintf(stderr, "shim_sao.ovg: two rect entries on two OVCTUDecs: first taken, second refused (%d, %d)\n", codes[0], codes[1]);
    }
    gfile_close(&g);
}


/* ====================================================================================== ALF */
static void
rand_alf_aps(OVALFData *a)
{
    memset(a, 0, sizeof(*a));
    int nf = rnd_range(1, 25);
    a->alf_luma_num_filters_signalled_minus1 = nf - 1;
    a->alf_luma_clip_flag = rnd_range(0, 1);
    for (int c = 0; c < 25; ++c) a->alf_luma_coeff_delta_idx[c] = rnd_range(0, nf - 1);
    for (int f = 0; f < 25; ++f) for (int k = 0; k < 12; ++k) {
        a->alf_luma_coeff[f][k] = (int16_t)(rnd_range(0, 7) == 0 ? rnd_range(-127, 127) : rnd_range(-24, 24));
        a->alf_luma_clip_idx[f][k] = rnd_range(0, 3);
    }
    a->alf_chroma_clip_flag = rnd_range(0, 1);
    a->alf_chroma_num_alt_filters_minus1 = rnd_range(0, 7);
    for (int f = 0; f < 8; ++f) for (int k = 0; k < 6; ++k) {
        a->alf_chroma_coeff[f][k] = (int16_t)(rnd_range(0, 7) == 0 ? rnd_range(-127, 127) : rnd_range(-24, 24));
        a->alf_chroma_clip_idx[f][k] = rnd_range(0, 3);
    }
    a->alf_cc_cb_filters_signalled_minus1 = 3; a->alf_cc_cr_filters_signalled_minus1 = 3;
    for (int c = 0; c < 2; ++c) for (int f = 0; f < 4; ++f) for (int k = 0; k < 7; ++k) {
        int m = rnd_range(0, 7);
        int v = m == 0 ? 0 : 1 << (m - 1);
        a->alf_cc_mapped_coeff[c][f][k] = (int16_t)(rnd_range(0, 1) ? -v : v);
    }
}

static void
gen_alf(const char *dir)
{
    enum { NPIC = 3 };
    static const int PW[NPIC] = { 304, 264, 136 }, PH[NPIC] = { 200, 256, 72 };
    gfile g = gfile_open(dir, g_shim ? "shim_alf.ovg" : "alf.ovg");
    g_seed = 0x266 + 4;
    for (int pi = 0; pi < NPIC; ++pi) {
        const int W = PW[pi], H = PH[pi], nx = (W + 127) / 128, ny = (H + 127) / 128;
        OVFrame *f = harness_frame(W, H);
        uint16_t *py = f->data[0];
        for (int i = 0; i < W * H; i += 13) py[i] = (uint16_t)rnd_range(0, 1023);
        char nm[32];
        uint32_t d2[2] = { H, W };
        if (!g_shim) {
        snprintf(nm, 32, "p%d_in_y", pi); gfile_array(&g, nm, T_U16, f->data[0], 2, d2);
        d2[0] = H / 2; d2[1] = W / 2;
        snprintf(nm, 32, "p%d_in_cb", pi); gfile_array(&g, nm, T_U16, f->data[1], 2, d2);
        snprintf(nm, 32, "p%d_in_cr", pi); gfile_array(&g, nm, T_U16, f->data[2], 2, d2);
        }

        OVCTUDec *c = ref_new_ctudec(0, 0);
        if (g_shim) shim_bind(c, W, H, 0, 0);
        c->pic_w = W; c->pic_h = H;
        c->rcn_ctx.frame_start = f;
        harness_alloc_filter_buffers(&c->rcn_ctx, nx, 3, 7);
        struct ALFInfo *ai = &c->alf_info;
        static OVALFData aps[5];
        for (int i = 0; i < 5; ++i) rand_alf_aps(&aps[i]);
        ai->alf_luma_enabled_flag = ai->alf_cb_enabled_flag = ai->alf_cr_enabled_flag = 1;
        ai->cc_alf_cb_enabled_flag = ai->cc_alf_cr_enabled_flag = 1;
        ai->num_alf_aps_ids_luma = 2;
        ai->aps_alf_data[0] = &aps[0]; ai->aps_alf_data[1] = &aps[1];
        ai->aps_alf_data_c = &aps[2]; ai->aps_cc_alf_data_cb = &aps[3]; ai->aps_cc_alf_data_cr = &aps[4];
        ai->ctb_alf_params = calloc(nx * ny, sizeof(ALFParamsCtu));
        ai->ctb_cc_alf_filter_idx[0] = calloc(nx * ny, 1); ai->ctb_cc_alf_filter_idx[1] = calloc(nx * ny, 1);
        c->rcn_funcs.alf.rcn_alf_reconstruct_coeff_APS(&ai->rcn_alf, c, 1, 1);
        ovhip_alf_ctu *mine = calloc(nx * ny, sizeof(*mine));
        for (int i = 0; i < nx * ny; ++i) {
            ALFParamsCtu *p = &ai->ctb_alf_params[i];
            p->ctb_alf_flag = rnd_range(0, 9) < 8 ? (rnd_range(0, 7) | (rnd_range(0, 2) ? 4 : 0)) : 0;
            p->ctb_alf_idx = rnd_range(0, 17);
            p->cb_alternative = rnd_range(0, aps[2].alf_chroma_num_alt_filters_minus1);
            p->cr_alternative = rnd_range(0, aps[2].alf_chroma_num_alt_filters_minus1);
            ai->ctb_cc_alf_filter_idx[0][i] = rnd_range(0, 4);
            ai->ctb_cc_alf_filter_idx[1][i] = rnd_range(0, 4);
            mine[i].flags = p->ctb_alf_flag; mine[i].luma_set = p->ctb_alf_idx;
            mine[i].cb_alt = p->cb_alternative; mine[i].cr_alt = p->cr_alternative;
            mine[i].cc_cb_idx = ai->ctb_cc_alf_filter_idx[0][i]; mine[i].cc_cr_idx = ai->ctb_cc_alf_filter_idx[1][i];
        }
        struct RectEntryInfo einfo;
        memset(&einfo, 0, sizeof(einfo));
        einfo.nb_ctu_w = nx; einfo.nb_ctu_h = ny;
        for (int cy = 0; cy < ny; ++cy) { c->ctb_y = cy; TIMED(TS_ALF, c->rcn_funcs.alf.rcn_alf_filter_line(c, &einfo, cy)); }

        if (g_shim) {
            size_t nc = 0, nt = 0;
            const ovhip_alf_ctu *ap = ovhip_shim_alf_params(c, &nc);
            if (!ap || nc != (size_t)(nx * ny) || ovhip_shim_last_error(c)) { fprintf(stderr, "shim: alf picture %d: no parameters captured\n", pi); exit(1); }
            d2[0] = nx * ny; d2[1] = sizeof(ovhip_alf_ctu);
            snprintf(nm, 32, "p%d_ctus", pi); gfile_array(&g, nm, T_U8, ap, 2, d2);
            static const char *tn[5] = { "luma_coeff", "luma_clip", "chroma_coeff", "chroma_clip", "cc_coeff" };
            for (int t = 0; t < 5; ++t) {
                const int16_t *tp = ovhip_shim_alf_table(c, t, &nt);
                uint32_t dt = (uint32_t)nt;
                snprintf(nm, 32, "p%d_%s", pi, tn[t]); gfile_array(&g, nm, T_I16, tp, 1, &dt);
            }
            continue;
        }
        d2[0] = H; d2[1] = W;
        snprintf(nm, 32, "p%d_exp_y", pi); gfile_array(&g, nm, T_U16, f->data[0], 2, d2);
        d2[0] = H / 2; d2[1] = W / 2;
        snprintf(nm, 32, "p%d_exp_cb", pi); gfile_array(&g, nm, T_U16, f->data[1], 2, d2);
        snprintf(nm, 32, "p%d_exp_cr", pi); gfile_array(&g, nm, T_U16, f->data[2], 2, d2);
        d2[0] = nx * ny; d2[1] = sizeof(ovhip_alf_ctu);
        snprintf(nm, 32, "p%d_ctus", pi); gfile_array(&g, nm, T_U8, mine, 2, d2);
        d2[0] = 24; d2[1] = OVHIP_ALF_LUMA_SET_SIZE;
        snprintf(nm, 32, "p%d_luma_coeff", pi); gfile_array(&g, nm, T_I16, ai->rcn_alf.filter_coeff_dec, 2, d2);
        snprintf(nm, 32, "p%d_luma_clip", pi); gfile_array(&g, nm, T_I16, ai->rcn_alf.filter_clip_dec, 2, d2);
        d2[0] = 8; d2[1] = 7;
        snprintf(nm, 32, "p%d_chroma_coeff", pi); gfile_array(&g, nm, T_I16, ai->rcn_alf.chroma_coeff_final, 2, d2);
        snprintf(nm, 32, "p%d_chroma_clip", pi); gfile_array(&g, nm, T_I16, ai->rcn_alf.chroma_clip_final, 2, d2);
        int16_t cc[2][4][8];
        memcpy(cc[0], aps[3].alf_cc_mapped_coeff[0], sizeof(cc[0]));
        memcpy(cc[1], aps[4].alf_cc_mapped_coeff[1], sizeof(cc[1]));
        uint32_t d3[3] = { 2, 4, 8 };
        snprintf(nm, 32, "p%d_cc_coeff", pi); gfile_array(&g, nm, T_I16, cc, 3, d3);
        fprintf(stderr, "alf.ovg: picture %d %dx%d (%d CTUs)\n", pi, W, H, nx * ny);
    }
    gfile_close(&g);
}

/* ====================================================================================== INTRA
 * intra.ovg : the reference's intra prediction slots on one CTU of a 2x2-CTU neighbourhood:
 *   intra_pred (planar / DC / 65 angular incl. wide angles, reference smoothing, fC / fG, PDPC, luma BDPCM),
 *   intra_pred_mrl, mip.rcn_intra_mip (+ transposed), intra_pred_c (planar / DC / angular / BDPCM) and cclm.{cclm,
 *   mdlm_left, mdlm_top} behind it                                  (rcn_structures.h:507-524, :558-593)
 * with every neighbour-availability pattern the progress bit-fields can express.  Inputs in the vocabulary of
 * include/ovvc_hip.h (ovhip_itask on a picture), outputs = the predicted block(s).
 * Picture = what the slots see of the CTU scratch (rcn_ctu.c:553-568, ctb_x = 1): the current CTU at luma (128, 128),
 * the previous CTU's columns to its left, one reconstructed row above, 64 columns of the CTU to the right. */
#define IN_W 328
#define IN_H 264
#define IN_OX 128
#define IN_OY 128

static void
gen_intra(const char *dir)
{
    gbuf b_task = { .type = T_U8 }, b_eoff = { .type = T_U32 }, b_exp = { .type = T_U16 };
    uint32_t n_cases = 0;
    g_seed = 0x266 + 555;
    OVCTUDec *c = ref_new_ctudec(0, 0);
    c->rcn_funcs.rcn_attach_ctu_buff(&c->rcn_ctx, 7, 1);
    const struct OVBuffInfo *cb = &c->rcn_ctx.ctu_buff;
    struct OVRCNCtx *r = &c->rcn_ctx;
    static uint16_t py[IN_W * IN_H], pcb[(IN_W / 2) * (IN_H / 2)], pcr[(IN_W / 2) * (IN_H / 2)];
    fill_plane(py, IN_W, IN_H, IN_W); fill_plane(pcb, IN_W / 2, IN_H / 2, IN_W / 2); fill_plane(pcr, IN_W / 2, IN_H / 2, IN_W / 2);
    for (int i = 0; i < IN_W * IN_H; i += 41) py[i] = (i & 1) ? 1023 : 0;
    /* CTU scratch <- picture: rows -1 .. 131, columns -128 .. 195 (what the buffer holds around the CTU) */
    #define LOAD_SCRATCH() do { \
        for (int j = -1; j < 132; ++j) for (int i = -128; i < 196; ++i) cb->y[j * cb->stride + i] = py[(IN_OY + j) * IN_W + IN_OX + i]; \
        for (int j = -1; j < 66; ++j) for (int i = -64; i < 98; ++i) { \
            cb->cb[j * cb->stride_c + i] = pcb[(IN_OY / 2 + j) * (IN_W / 2) + IN_OX / 2 + i]; \
            cb->cr[j * cb->stride_c + i] = pcr[(IN_OY / 2 + j) * (IN_W / 2) + IN_OX / 2 + i]; } } while (0)

    /* kind: 0 luma regular, 1 luma MRL, 2 MIP, 3 chroma regular, 4 chroma LM, 5 luma BDPCM, 6 chroma BDPCM */
    for (int kind = 0; kind < 7; ++kind) {
        const int chroma = kind == 3 || kind == 4 || kind == 6;
        const int lmin = chroma ? 1 : 2, lmax = chroma ? 5 : 6;
        for (int l2w = lmin; l2w <= lmax; ++l2w) {
            for (int l2h = lmin; l2h <= lmax; ++l2h) {
                if (chroma && l2w + l2h < 3) continue;                   /* no 2x2 chroma blocks */
                if (kind == 2 && (l2w > 6 || l2h > 6)) continue;
                if ((kind == 5 || kind == 6) && (l2w > 5 || l2h > 5)) continue;
                const int w = 1 << l2w, h = 1 << l2h, unit = chroma ? 2 : 4, ctu = chroma ? 64 : 128;
                int n_modes = kind == 0 || kind == 3 ? 67 : kind == 1 ? 67 : kind == 2 ? 32 : kind == 4 ? 3 : 2;
                int reps = kind == 0 ? (w * h <= 256 ? 3 : 2) : kind == 4 ? (w * h >= 256 ? 3 : 6) : 1;
                for (int mi = 0; mi < n_modes; ++mi) {
                    /* keep the fixture small: big blocks take every third (fifth) mode plus the structurally special ones */
                    const int key_mode = mi <= 2 || mi == 18 || mi == 34 || mi == 50 || mi == 66;
                    const int thin = w * h >= 2048 ? 5 : w * h >= 512 ? 3 : (kind == 1 && w * h >= 128 ? 2 : 1);
                    if ((kind == 0 || kind == 1 || kind == 3) && !key_mode && (mi % thin) != ((l2w * 3 + l2h) % thin)) continue;
                    if (kind == 2 && w * h >= 1024 && (mi & 1) != ((l2w + l2h) & 1)) continue;
                    for (int rep = 0; rep < (w * h >= 512 && kind != 4 ? 1 : reps); ++rep) {
                        ovhip_itask t;
                        memset(&t, 0, sizeof(t));
                        int mode = mi, mrl = 0, mip_tr = 0;
                        if (kind == 1) { if (mode == 0 || mode == 1) { if (rep) continue; } mrl = 1 + ((mi + rep) & 1); }
                        if (kind == 2) {
                            const int n_mip = (l2w == 2 && l2h == 2) ? 16 : (l2h == 2 || l2w == 2 || (l2h <= 3 && l2w <= 3)) ? 8 : 6;
                            mode = mi % n_mip; mip_tr = (mi / n_mip) & 1;
                            if (mi >= 2 * n_mip) continue;
                        }
                        if (kind == 4) mode = 67 + mi;
                        /* position inside the CTU, availability pattern */
                        int x0, y0;
                        x0 = rnd_range(0, (ctu - w) / unit) * unit; y0 = rnd_range(0, (ctu - h) / unit) * unit;
                        if (rep % 3 == 1 || (kind != 0 && rnd_range(0, 3) == 0)) { if (rnd_range(0, 1)) x0 = 0; else y0 = 0; }
                        if (kind == 1 && y0 < 4) y0 = 4 * rnd_range(1, (ctu - h) / 4 > 1 ? (ctu - h) / 4 : 1);
                        if (kind == 1 && y0 + h > ctu) continue;
                        int max_abv = 2 * w / unit, max_lft = 2 * h / unit;
                        int cap_abv = ((chroma ? 96 : 192) - x0) / unit, cap_lft = ((chroma ? 64 : 128) - y0) / unit;
                        if (max_abv > cap_abv) max_abv = cap_abv;
                        if (max_lft > cap_lft) max_lft = cap_lft;
                        int corner = 1, avl_abv = max_abv, avl_lft = max_lft;
                        /* the availability states a decoder can be in (decoding order + slice / picture borders) */
                        int pat = rnd_range(0, 9);
                        if (kind == 1 && pat < 3) pat += 3;                                       /* MRL is not signalled on the first CTU row */
                        if (pat == 0) { corner = 0; avl_abv = 0; avl_lft = 0; }                  /* nothing (picture / slice corner) */
                        else if (pat == 1) { corner = 0; avl_abv = 0; }                           /* top row */
                        else if (pat == 2) { corner = 0; avl_lft = 0; }                           /* left column */
                        else if (pat <= 5) { avl_abv = rnd_range(w / unit < max_abv ? w / unit : max_abv, max_abv);
                                             avl_lft = rnd_range(h / unit < max_lft ? h / unit : max_lft, max_lft); }
                        else if (pat == 6) { avl_abv = w / unit < max_abv ? w / unit : max_abv; avl_lft = h / unit < max_lft ? h / unit : max_lft; }
                        else if (pat == 7 && kind != 1) { corner = 0; }                          /* a slice starts at the CTU above */
                        /* progress bit-fields: bit (unit + 1) of hfield[row above] / vfield[column left]; bit `unit` = the corner */
                        struct CTUBitField *pf = chroma ? &r->progress_field_c : &r->progress_field;
                        memset(pf, 0, sizeof(*pf));
                        const int xu = x0 / unit, yu = y0 / unit;
                        pf->hfield[yu] = (((uint64_t)corner) | ((((uint64_t)1 << avl_abv) - 1) << 1)) << xu;
                        pf->vfield[xu] = (((uint64_t)corner) | ((((uint64_t)1 << avl_lft) - 1) << 1)) << yu;
                        LOAD_SCRATCH();
                        CUFlags fl = flg_pred_mode_flag;
                        t.x = (uint16_t)((chroma ? IN_OX / 2 : IN_OX) + x0); t.y = (uint16_t)((chroma ? IN_OY / 2 : IN_OY) + y0);
                        t.log2_w = l2w; t.log2_h = l2h; t.kind = chroma ? OVHIP_IT_CHROMA : OVHIP_IT_LUMA;
                        t.mode = (uint8_t)mode; t.flags = corner ? OVHIP_IF_CORNER : 0;
                        t.avl_lft = avl_lft; t.avl_abv = avl_abv; t.mrl_idx = mrl; t.level = 1;
                        const double t_in = g_time ? now_s() : 0.0;
                        switch (kind) {
                        case 0: c->rcn_funcs.intra_pred(r, cb, mode, x0, y0, l2w, l2h, fl); break;
                        case 1: c->rcn_funcs.intra_pred_mrl(c, cb->y, cb->stride, mode, x0, y0, l2w, l2h, mrl); break;
                        case 2: t.flags |= OVHIP_IF_MIP | (mip_tr ? OVHIP_IF_MIP_TR : 0);
                                c->rcn_funcs.mip.rcn_intra_mip(r, x0, y0, l2w, l2h, (uint8_t)(mode | (mip_tr << 7))); break;
                        case 5: fl |= flg_intra_bdpcm_luma_flag | (mi ? flg_intra_bdpcm_luma_dir : 0);
                                t.flags |= OVHIP_IF_BDPCM | (mi ? OVHIP_IF_BDPCM_VER : 0); t.mode = 0;
                                c->rcn_funcs.intra_pred(r, cb, 0, x0, y0, l2w, l2h, fl); break;
                        case 6: fl |= flg_intra_bdpcm_chroma_flag | (mi ? flg_intra_bdpcm_chroma_dir : 0);
                                t.flags |= OVHIP_IF_BDPCM | (mi ? OVHIP_IF_BDPCM_VER : 0); t.mode = 0;
                                c->rcn_funcs.intra_pred_c(r, 0, x0, y0, l2w, l2h, fl); break;
                        case 4: {
                            /* the LM modes read their own availability: abv / lft "any unit" flags, MDLM: contiguous units
                             * over w + min(w, h) (h + min(w, h)) samples (rcn_intra_cclm.c:56-68, :770-776, :843-849) */
                            const int any_abv = avl_abv > 0, any_lft = avl_lft > 0;
                            int need_a = (w + (w < h ? w : h)) / 2, need_l = (h + (w < h ? w : h)) / 2;
                            t.avl_abv = mode == 69 ? (avl_abv < need_a ? avl_abv : need_a) : any_abv;
                            t.avl_lft = mode == 68 ? (avl_lft < need_l ? avl_lft : need_l) : any_lft;
                            c->rcn_funcs.intra_pred_c(r, mode, x0, y0, l2w, l2h, fl); break; }
                        default: c->rcn_funcs.intra_pred_c(r, mode, x0, y0, l2w, l2h, fl); break;
                        }
                        if (g_time) g_ts[TS_INTRA] += now_s() - t_in;
                        uint32_t eoff[2] = { 0, 0 };
                        if (!chroma) { eoff[0] = (uint32_t)b_exp.n; dump_rect(&b_exp, cb->y, cb->stride, x0, y0, w, h); }
                        else { eoff[0] = (uint32_t)b_exp.n; dump_rect(&b_exp, cb->cb, cb->stride_c, x0, y0, w, h);
                               eoff[1] = (uint32_t)b_exp.n; dump_rect(&b_exp, cb->cr, cb->stride_c, x0, y0, w, h); }
                        gbuf_push(&b_task, &t, sizeof(t)); gbuf_push(&b_eoff, eoff, 2);
                        n_cases++;
                    }
                }
            }
        }
    }
    gfile g = gfile_open(dir, "intra.ovg");
    uint32_t d2[2] = { IN_H, IN_W };
    gfile_array(&g, "pic_y", T_U16, py, 2, d2);
    d2[0] = IN_H / 2; d2[1] = IN_W / 2;
    gfile_array(&g, "pic_cb", T_U16, pcb, 2, d2); gfile_array(&g, "pic_cr", T_U16, pcr, 2, d2);
    d2[0] = n_cases; d2[1] = sizeof(ovhip_itask); gfile_array(&g, "task", T_U8, b_task.data, 2, d2);
    d2[1] = 2; gfile_array(&g, "exp_off", T_U32, b_eoff.data, 2, d2);
    gfile_buf(&g, "exp", &b_exp);
    gfile_close(&g);
    fprintf(stderr, "intra.ovg: %u cases, %zu expected samples\n", n_cases, b_exp.n);
}

/* ---------------------------------------------------------------------------------------------------------------------
 * K12  whole intra CTUs through tmp.rcn_transform_tree (rcn_transform_tree.c:1454-1518 -> rcn_res_wrap -> rcn_intra_tu /
 *      rcn_tu_st / rcn_tu_l / rcn_tu_c): a random partition of the CTU at (128, 128), CUs in decoding order, every CU intra
 *      (regular / MRL / MIP / BDPCM luma, DC / planar / angular / DM / LM / MDLM chroma, with residuals: DCT-2, implicit MTS,
 *      LFNST, transform skip) or left as it is ("decoded before", as an inter CU would be).  Single tree and dual tree.
 *      Reference mode: intra_ctu.ovg = the start picture + the CTU each case ends with.
 *      Shim mode: shim_intra_ctu.ovg = what the installed slots recorded for the same CTUs (commands, coefficients, ordered
 *      tasks).  Executing the stream on the start picture must give the reference's CTU. */
struct ictu_cu { int x, y, l2w, l2h; };
static void
ictu_part(struct ictu_cu *out, int *n, int x, int y, int l2w, int l2h)
{
    const int w = 1 << l2w, h = 1 << l2h;
    int choice = 0;                                  /* 0 leaf, 1 quad, 2 vertical, 3 horizontal */
    if (w > 64 || h > 64) choice = 1;
    else {
        const int r = rnd_range(0, 99);
        const int p_leaf = w * h >= 2048 ? 25 : w * h >= 512 ? 45 : w * h >= 128 ? 60 : 85;
        if (r >= p_leaf) {
            int opts[3], no = 0;
            if (l2w == l2h && l2w >= 4) opts[no++] = 1;
            if (l2w >= 4) opts[no++] = 2;
            if (l2h >= 3) opts[no++] = 3;
            if (no) choice = opts[rnd_range(0, no - 1)];
        }
    }
    if (choice == 0) { out[*n] = (struct ictu_cu){ x, y, l2w, l2h }; ++*n; return; }
    if (choice == 1) {
        ictu_part(out, n, x, y, l2w - 1, l2h - 1); ictu_part(out, n, x + w / 2, y, l2w - 1, l2h - 1);
        ictu_part(out, n, x, y + h / 2, l2w - 1, l2h - 1); ictu_part(out, n, x + w / 2, y + h / 2, l2w - 1, l2h - 1);
    } else if (choice == 2) { ictu_part(out, n, x, y, l2w - 1, l2h); ictu_part(out, n, x + w / 2, y, l2w - 1, l2h); }
    else { ictu_part(out, n, x, y, l2w, l2h - 1); ictu_part(out, n, x, y + h / 2, l2w, l2h - 1); }
}

static void
gen_intra_ctu(const char *dir)
{
    extern int transform_unit_st(OVCTUDec *const, unsigned int, unsigned int, unsigned int, unsigned int, uint8_t, CUFlags, uint8_t, struct TUInfo *const);
    extern int transform_unit_l(OVCTUDec *const, unsigned int, unsigned int, unsigned int, unsigned int, uint8_t, CUFlags, uint8_t, struct TUInfo *const);
    extern int transform_unit_c(OVCTUDec *const, unsigned int, unsigned int, unsigned int, unsigned int, uint8_t, CUFlags, uint8_t, struct TUInfo *const);
    gbuf b_exp = { .type = T_U16 }, b_info = { .type = T_I32 };
    uint32_t n_cases = 0;
    g_seed = 0x266 + 777;
    OVCTUDec *c = ref_new_ctudec(0, 0);
    c->rcn_funcs.rcn_attach_ctu_buff(&c->rcn_ctx, 7, 1);
    const struct OVBuffInfo *cb = &c->rcn_ctx.ctu_buff;
    struct OVRCNCtx *r = &c->rcn_ctx;
    static uint16_t py[IN_W * IN_H], pcb[(IN_W / 2) * (IN_H / 2)], pcr[(IN_W / 2) * (IN_H / 2)];
    fill_plane(py, IN_W, IN_H, IN_W); fill_plane(pcb, IN_W / 2, IN_H / 2, IN_W / 2); fill_plane(pcr, IN_W / 2, IN_H / 2, IN_W / 2);
    struct shim_stream S;
    shim_stream_init(&S);
    c->ctb_x = IN_OX >> 7; c->ctb_y = IN_OY >> 7;
    if (g_shim) shim_bind(c, IN_W, IN_H, 0, 0);

    for (int ci = 0; ci < 36; ++ci) {
        const int dual = ci % 3 == 2;
        struct ictu_cu cus[1024]; int n_cu = 0;
        ictu_part(cus, &n_cu, 0, 0, 7, 7);
        const int ict_type = rnd_range(0, 3);
        rcn_init_ict_functions_10(&c->rcn_funcs, ict_type, 10);
        if (g_shim) rcn_init_functions_hip(&c->rcn_funcs, ict_type, 1, 0, 0, 10);
        c->dequant_luma.qp = rnd_range(18, 50); c->dequant_cb.qp = rnd_range(18, 50); c->dequant_cr.qp = rnd_range(18, 50);
        c->dequant_joint_cb_cr.qp = rnd_range(18, 50);
        c->dequant_luma_skip.qp = c->dequant_luma.qp; c->dequant_cb_skip.qp = c->dequant_cb.qp; c->dequant_cr_skip.qp = c->dequant_cr.qp;
        c->dequant_jcbcr_skip.qp = c->dequant_joint_cb_cr.qp;
        c->residual_coding_l = rnd_range(0, 1) ? &residual_coding_dpq : NULL;
        c->mts_implicit = rnd_range(0, 1); c->sh_ts_disabled = 0; c->tmp_ciip = 0;
        c->lmcs_info.scale_c_flag = rnd_range(0, 1); c->lmcs_info.lmcs_chroma_scale = (uint16_t)rnd_range(1200, 3400);
        memset(&c->dbf_info, 0, sizeof(c->dbf_info));
        /* the CTU scratch holds the picture around the CTU; the neighbours that exist: left and above CTUs, and the part of the
         * above-right one inside the picture */
        for (int j = -1; j < 132; ++j) for (int i = -128; i < 196; ++i) cb->y[j * cb->stride + i] = py[(IN_OY + j) * IN_W + IN_OX + i];
        for (int j = -1; j < 66; ++j) for (int i = -64; i < 98; ++i) {
            cb->cb[j * cb->stride_c + i] = pcb[(IN_OY / 2 + j) * (IN_W / 2) + IN_OX / 2 + i];
            cb->cr[j * cb->stride_c + i] = pcr[(IN_OY / 2 + j) * (IN_W / 2) + IN_OX / 2 + i];
        }
        /* the decoder's own CTU start (decode_ctu, slicedec.c:731-737): left, above and above-right CTUs exist, the last one
         * only as far as the picture goes */
        memset(&r->progress_field, 0, sizeof(r->progress_field)); memset(&r->progress_field_c, 0, sizeof(r->progress_field_c));
        init_ctu_bitfield(r, CTU_LFT_FLG | CTU_UP_FLG | CTU_UPRGT_FLG, 7);
        {
            const uint64_t mask = ((uint64_t)1 << ((((IN_W - IN_OX - 128) + 128) >> 2) + 1)) - 1;
            r->progress_field_c.hfield[0] &= mask; r->progress_field.hfield[0] &= mask;
        }
        int n_intra = 0;
        uint8_t luma_mode_of[1024];
        for (int pass = 0; pass < (dual ? 2 : 1); ++pass) {
            c->transform_unit = dual ? (pass ? (void *)&transform_unit_c : (void *)&transform_unit_l) : (void *)&transform_unit_st;
            for (int k = 0; k < n_cu; ++k) {
                const int x0 = cus[k].x, y0 = cus[k].y, l2w = cus[k].l2w, l2h = cus[k].l2h, w = 1 << l2w, h = 1 << l2h;
                struct TUInfo tu;
                memset(&tu, 0, sizeof(tu));
                /* "decoded before": the same CUs in both passes of a dual tree */

                const int skip = ((cus[k].x * 31 + cus[k].y * 17 + ci * 7) % 100) < 22;
                if (skip) {
                    struct CTUBitField *pf = (dual && pass) ? &r->progress_field_c : &r->progress_field;
                    ctu_field_set_rect_bitfield(pf, x0 >> 2, y0 >> 2, w >> 2, h >> 2);
                    if (!dual) ctu_field_set_rect_bitfield(&r->progress_field_c, x0 >> 2, y0 >> 2, w >> 2, h >> 2);
                    continue;
                }
                CUFlags fl = flg_pred_mode_flag;
                int kind = rnd_range(0, 99);                      /* luma: regular / MRL / MIP / BDPCM */
                kind = kind < 50 ? 0 : kind < 65 ? 1 : kind < 82 ? 2 : 3;
                if (kind == 1 && y0 == 0) kind = 0;               /* MRL is not signalled on the first line of a CTU */
                if (kind == 3 && (l2w > 5 || l2h > 5)) kind = 0;
                int mode = rnd_range(0, 66), mode_c;
                int bdpcm_c = 0;
                if (!(dual && pass)) {
                    c->cu_opaque = 0;
                    if (kind == 1) { fl |= flg_mrl_flag; c->cu_opaque = (uint8_t)rnd_range(1, 2); }
                    if (kind == 2) {
                        const int n_mip = (l2w == 2 && l2h == 2) ? 16 : (l2h == 2 || l2w == 2 || (l2h <= 3 && l2w <= 3)) ? 8 : 6;
                        fl |= flg_mip_flag; c->cu_opaque = (uint8_t)(rnd_range(0, n_mip - 1) | (rnd_range(0, 1) << 7));
                        mode = 0;
                    }
                    if (kind == 3) { fl |= flg_intra_bdpcm_luma_flag | (rnd_range(0, 1) ? flg_intra_bdpcm_luma_dir : 0); mode = (fl & flg_intra_bdpcm_luma_dir) ? 50 : 18; }
                    luma_mode_of[k] = (uint8_t)mode;
                } else {
                    mode = luma_mode_of[k];
                }
                c->intra_mode = (uint8_t)mode;
                {
                    static const uint8_t cm[8] = { 0, 1, 18, 50, 255, 67, 68, 69 };
                    mode_c = cm[rnd_range(0, 7)];
                    if (mode_c == 255) mode_c = mode;             /* DM */
                    if (kind == 3 && l2w <= 5 && l2h <= 5 && rnd_range(0, 1)) { bdpcm_c = 1; fl |= flg_intra_bdpcm_chroma_flag | (rnd_range(0, 1) ? flg_intra_bdpcm_chroma_dir : 0); }
                }
                c->intra_mode_c = (uint8_t)mode_c;
                /* residual */
                const int has_l = !(dual && pass) && rnd_range(0, 99) < 70;
                static const uint8_t cbf_c[7] = { 0, 0x2, 0x1, 0x3, 0xb, 0xa, 0x9 };
                int cbfc = (dual && !pass) ? 0 : cbf_c[rnd_range(0, 6)];
                if (bdpcm_c) cbfc &= 0x3;
                tu.cbf_mask = (uint8_t)((has_l ? 0x10 : 0) | cbfc);
                int lfnst = !(fl & (flg_intra_bdpcm_luma_flag)) && has_l && rnd_range(0, 3) == 0;
                if (lfnst) { tu.lfnst_flag = 1; tu.lfnst_idx = (uint8_t)rnd_range(0, 1); }
                if (fl & flg_intra_bdpcm_luma_flag) tu.tr_skip_mask |= 0x10;
                if (bdpcm_c) tu.tr_skip_mask |= 0x3;
                int16_t *res[3] = { c->residual_cb, c->residual_cr, c->residual_y };
                const int cl2w = (dual && pass) ? l2w - 1 : l2w - 1, cl2h = l2h - 1;
                for (int comp = 0; comp < 3; ++comp) {
                    const int is_l = comp == 2;
                    const int used = is_l ? has_l : ((cbfc & 0x8) ? comp == 0 : !!(cbfc & (comp ? 0x1 : 0x2)));
                    if (!used) continue;
                    const int tl2w = is_l ? l2w : cl2w, tl2h = is_l ? l2h : cl2h;
                    const int ts = is_l ? !!(tu.tr_skip_mask & 0x10) : (cbfc & 0x8) ? !!(tu.tr_skip_mask & 0x1) : !!(tu.tr_skip_mask & (comp ? 0x1 : 0x2));
                    const int raster = ts || tl2w < 2 || tl2h < 2;
                    uint64_t map; uint16_t lp;
                    make_coefs(res[comp], tl2w, tl2h, (is_l && lfnst) ? 1 : rnd_range(0, 2), raster, &map, &lp);
                    if (is_l && lfnst) {
                        map = 1; lp = 0x0101;
                        if (!raster) { int cw = (1 << tl2w) > 32 ? 32 : (1 << tl2w), ch = (1 << tl2h) > 32 ? 32 : (1 << tl2h); memset(res[comp] + 16, 0, (cw * ch - 16) * 2); }
                    }
                    tu.tb_info[comp].sig_sb_map = map; tu.tb_info[comp].last_pos = lp;
                }
                if (dual && pass) c->rcn_funcs.tmp.rcn_transform_tree(c, x0 >> 1, y0 >> 1, l2w - 1, l2h - 1, 5, 0, fl, &tu);
                else c->rcn_funcs.tmp.rcn_transform_tree(c, x0, y0, l2w, l2h, 6, 0, fl, &tu);
                ++n_intra;
            }
        }
        if (g_shim) shim_case_end(c, &S, "intra ctu");
        int32_t info[4] = { dual, n_cu, n_intra, (int32_t)b_exp.n };
        gbuf_push(&b_info, info, 4);
        dump_rect(&b_exp, cb->y, cb->stride, 0, 0, 128, 128);
        dump_rect(&b_exp, cb->cb, cb->stride_c, 0, 0, 64, 64);
        dump_rect(&b_exp, cb->cr, cb->stride_c, 0, 0, 64, 64);
        n_cases++;
    }
    if (g_shim) { shim_stream_write(dir, "shim_intra_ctu.ovg", &S, c, NULL, 0); return; }
    gfile g = gfile_open(dir, "intra_ctu.ovg");
    uint32_t d2[2] = { IN_H, IN_W };
    gfile_array(&g, "pic_y", T_U16, py, 2, d2);
    d2[0] = IN_H / 2; d2[1] = IN_W / 2;
    gfile_array(&g, "pic_cb", T_U16, pcb, 2, d2); gfile_array(&g, "pic_cr", T_U16, pcr, 2, d2);
    d2[0] = n_cases; d2[1] = 4; gfile_array(&g, "info", T_I32, b_info.data, 2, d2);
    gfile_buf(&g, "exp", &b_exp);
    gfile_close(&g);
    fprintf(stderr, "intra_ctu.ovg: %u CTUs, %zu expected samples\n", n_cases, b_exp.n);
}

/* ---------------------------------------------------------------------------------------------------------------------
 * K13  intra sub-partitions: tmp.recon_isp_subtree_v / _h (rcn_transform_tree.c:1087-1205 -> intra_pred_isp, rcn_isp_tu,
 *      rcn_2xX_tb / rcn_1xX_tb / rcn_Xx2_tb / rcn_Xx1_tb): every CU shape ISP exists for (4x8 ... 64x64), both split
 *      directions, planar / DC / angular modes incl. wide angles, DST-VII on and off, LFNST, random cbf patterns; the CU sits in
 *      a CTU whose surroundings are partly decoded.  isp.ovg = start picture + the CU each case ends with; shim mode:
 *      shim_isp.ovg = what the installed slots recorded. */
struct ISPTUInfo_h { uint8_t cbf_mask, tr_skip_mask, cu_mts_flag, cu_mts_idx, lfnst_flag, lfnst_idx; struct TBInfo tb_info[4]; };
static void
gen_isp(const char *dir)
{
    gbuf b_exp = { .type = T_U16 }, b_info = { .type = T_I32 };
    uint32_t n_cases = 0;
    g_seed = 0x266 + 999;
    OVCTUDec *c = ref_new_ctudec(0, 0);
    c->rcn_funcs.rcn_attach_ctu_buff(&c->rcn_ctx, 7, 1);
    const struct OVBuffInfo *cb = &c->rcn_ctx.ctu_buff;
    struct OVRCNCtx *r = &c->rcn_ctx;
    static uint16_t py[IN_W * IN_H], pcb[(IN_W / 2) * (IN_H / 2)], pcr[(IN_W / 2) * (IN_H / 2)];
    fill_plane(py, IN_W, IN_H, IN_W); fill_plane(pcb, IN_W / 2, IN_H / 2, IN_W / 2); fill_plane(pcr, IN_W / 2, IN_H / 2, IN_W / 2);
    struct shim_stream S;
    shim_stream_init(&S);
    c->ctb_x = IN_OX >> 7; c->ctb_y = IN_OY >> 7;
    if (g_shim) shim_bind(c, IN_W, IN_H, 0, 0);
    for (int l2w = 2; l2w <= 6; ++l2w)
        for (int l2h = 2; l2h <= 6; ++l2h) {
            if (l2w + l2h < 5) continue;                                 /* no ISP for 4x4 */
            for (int vertical = 0; vertical < 2; ++vertical) {
                /* the standard forbids a split that would leave partitions wider / higher than the maximum transform size only;
                 * both directions exist for every shape here */
                const int reps = (l2w + l2h <= 8) ? 10 : 6;
                /* 64x8 split horizontally = 64x2 partitions: rcn_Xx2_tb de-quantises with a row stride of 32 (dequant_tb, min(5,
                 * log2_tb_w)) but transforms with a stride of 64, reading stack memory nothing initialised (:985-1009): the
                 * reference's output for these is not a function of its inputs, there is nothing to pin (the recorder refuses them) */
                if (!vertical && l2w == 6 && l2h == 3) continue;
                for (int rep = 0; rep < reps; ++rep) {
                    const int w = 1 << l2w, h = 1 << l2h;
                    int32_t l2p, n_pb, l2pred, n_pred;
                    ovhip_isp_geometry(l2w, l2h, vertical, &l2p, &n_pb, &l2pred, &n_pred);
                    const int l2tw = vertical ? l2p : l2w, l2th = vertical ? l2h : l2p;
                    int x0 = rnd_range(0, (128 - w) / 4) * 4, y0 = rnd_range(0, (128 - h) / 4) * 4;
                    if (rep % 4 == 1) { if (rnd_range(0, 1)) x0 = 0; else y0 = 0; }
                    static const uint8_t key[8] = { 0, 1, 2, 18, 34, 50, 66, 0 };
                    const int mode = rep < 7 ? key[rep] : rnd_range(2, 66);
                    const int mode2 = (rep & 1) ? rnd_range(2, 66) : mode;
                    const uint8_t intra_mode = (uint8_t)(rep >= 3 && rep < 7 ? mode2 : mode);
                    c->dequant_luma.qp = rnd_range(18, 50); c->dequant_luma_skip.qp = c->dequant_luma.qp;
                    c->residual_coding_l = rnd_range(0, 1) ? &residual_coding_dpq : NULL;
                    c->mts_enabled = (uint8_t)rnd_range(0, 1); c->mts_implicit = 0; c->sh_ts_disabled = 0; c->tmp_ciip = 0;
                    c->intra_mode = intra_mode;
                    memset(&c->dbf_info, 0, sizeof(c->dbf_info));
                    for (int j = -1; j < 132; ++j) for (int i = -128; i < 196; ++i) cb->y[j * cb->stride + i] = py[(IN_OY + j) * IN_W + IN_OX + i];
                    /* progress: the decoder's CTU start, the rows above the CU and the columns left of it inside the CTU decoded
                     * (with some of the left part further down), then the CU itself (vcl_transform_unit.c:1878) */
                    memset(&r->progress_field, 0, sizeof(r->progress_field)); memset(&r->progress_field_c, 0, sizeof(r->progress_field_c));
                    int flags = CTU_LFT_FLG | CTU_UP_FLG | CTU_UPRGT_FLG;
                    if (rep % 5 == 2) flags &= ~CTU_UP_FLG & ~CTU_UPRGT_FLG;
                    if (rep % 5 == 3) flags &= ~CTU_LFT_FLG;
                    init_ctu_bitfield(r, (uint8_t)flags, 7);
                    { const uint64_t mask = ((uint64_t)1 << ((((IN_W - IN_OX - 128) + 128) >> 2) + 1)) - 1; r->progress_field.hfield[0] &= mask; }
                    const int xu = x0 >> 2, yu = y0 >> 2, wu = w >> 2, hu = h >> 2;
                    if (yu) ctu_field_set_rect_bitfield(&r->progress_field, 0, 0, rnd_range(0, 2) ? 32 : (xu + wu < 32 ? xu + wu : 32), yu);
                    if (xu) { int nl = rnd_range(hu, 2 * hu); if (yu + nl > 32) nl = 32 - yu; ctu_field_set_rect_bitfield(&r->progress_field, 0, yu, xu, nl); }
                    ctu_field_set_rect_bitfield(&r->progress_field, xu, yu, wu, hu);
                    /* residual */
                    struct ISPTUInfo_h tu;
                    memset(&tu, 0, sizeof(tu));
                    memset(c->residual_y, 0, sizeof(c->residual_y));
                    tu.cbf_mask = (uint8_t)rnd_range(1, (1 << n_pb) - 1);
                    const int can_lfnst = l2tw >= 2 && l2th >= 2;
                    if (can_lfnst && rnd_range(0, 3) == 0) { tu.lfnst_flag = 1; tu.lfnst_idx = (uint8_t)rnd_range(0, 1); }
                    for (int i = 0; i < n_pb; ++i) {
                        if (!((tu.cbf_mask >> (n_pb - i - 1)) & 1)) continue;
                        int16_t *dst = c->residual_y + (i << (l2tw + l2th));
                        uint64_t map; uint16_t lp;
                        if (l2tw >= 2 && l2th >= 2) {
                            make_coefs(dst, l2tw, l2th, tu.lfnst_flag ? 1 : rnd_range(0, 2), 0, &map, &lp);
                            if (tu.lfnst_flag) { map = 1; lp = 0x0101; int cw = (1 << l2tw) > 32 ? 32 : (1 << l2tw), ch = (1 << l2th) > 32 ? 32 : (1 << l2th); memset(dst + 16, 0, (cw * ch - 16) * 2); }
                        } else {
                            /* thin blocks: raster; the sub-block map of 2x8 / 8x2 (1x16 / 16x1) coefficient groups along the long side */
                            const int n = 1 << (l2tw + l2th);
                            const int pattern = rnd_range(0, 2);
                            for (int k = 0; k < n; ++k) dst[k] = (pattern == 2 || (pattern == 1 && rnd_range(0, 2) == 0) || k == 0) ? rnd_coef() : 0;
                            const int n_sb = n >> 4;                     /* 16 coefficients per group */
                            map = 0;
                            for (int sb = 0; sb < (n_sb ? n_sb : 1); ++sb) map |= (uint64_t)1 << (vertical ? sb * 8 : sb);
                            if (pattern == 0) { map = 1; for (int k = 1; k < n; ++k) dst[k] = 0; }
                            lp = (uint16_t)(pattern == 0 ? 0 : (vertical ? ((((1 << l2th) - 1) & 0x1f) << 8) | (((1 << l2th) - 1) & 0x1f) : ((1 << l2tw) - 1) & 0x1f));
                        }
                        tu.tb_info[i].sig_sb_map = map; tu.tb_info[i].last_pos = lp;
                    }
                    rcn_init_ict_functions_10(&c->rcn_funcs, 0, 10);
                    if (g_shim) rcn_init_functions_hip(&c->rcn_funcs, 0, 1, 0, 0, 10);
                    if (vertical) c->rcn_funcs.tmp.recon_isp_subtree_v(c, x0, y0, l2w, l2h, intra_mode, (const void *)&tu);
                    else          c->rcn_funcs.tmp.recon_isp_subtree_h(c, x0, y0, l2w, l2h, intra_mode, (const void *)&tu);
                    if (g_shim) shim_case_end(c, &S, "isp");
                    int32_t info[8] = { IN_OX + x0, IN_OY + y0, l2w, l2h, vertical, intra_mode, tu.cbf_mask | (tu.lfnst_flag << 8) | (c->mts_enabled << 9), (int32_t)b_exp.n };
                    gbuf_push(&b_info, info, 8);
                    dump_rect(&b_exp, cb->y, cb->stride, x0, y0, w, h);
                    n_cases++;
                }
            }
        }
    if (g_shim) { shim_stream_write(dir, "shim_isp.ovg", &S, c, NULL, 0); return; }
    gfile g = gfile_open(dir, "isp.ovg");
    uint32_t d2[2] = { IN_H, IN_W };
    gfile_array(&g, "pic_y", T_U16, py, 2, d2);
    d2[0] = n_cases; d2[1] = 8; gfile_array(&g, "info", T_I32, b_info.data, 2, d2);
    gfile_buf(&g, "exp", &b_exp);
    gfile_close(&g);
    fprintf(stderr, "isp.ovg: %u CUs, %zu expected samples\n", n_cases, b_exp.n);
}

int
main(int argc, char **argv)
{
    const char *dir = argc > 1 ? argv[1] : "../tests/golden";
    const char *only = argc > 2 ? argv[2] : NULL;
    if (only && !strcmp(only, "shim")) { g_shim = 1; only = argc > 3 ? argv[3] : NULL; }
    /* "simd": the same generators through the reference's SSE4.1 / AVX2 slots (`gen_golden <scratch dir> simd` rewrites the
     * fixtures from them -- byte-identical to the scalar ones if the reference's back-ends agree; `... simd time` times them) */
    if (only && !strcmp(only, "simd")) { g_simd = 1; only = argc > 3 ? argv[3] : NULL; }
    if (only && !strcmp(only, "time")) {
        /* five passes over the generators, the best time per stage; fixtures go to `dir` (use a scratch directory) */
        double best[TS_N];
        for (int k = 0; k < TS_N; ++k) best[k] = 1e30;
        g_time = 1;
        for (int pass = 0; pass < 5; ++pass) {
            memset(g_ts, 0, sizeof(g_ts));
            gen_itx(dir); gen_mc(dir); gen_dbf(dir); gen_sao(dir); gen_alf(dir); gen_intra(dir);
            for (int k = 0; k < TS_N; ++k) if (g_ts[k] > 0 && g_ts[k] < best[k]) best[k] = g_ts[k];
        }
        printf("{");
        int first = 1;
        for (int k = 0; k < TS_N; ++k) if (best[k] < 1e29) { printf("%s\"%s\": %.6f", first ? "" : ", ", g_ts_name[k], best[k]); first = 0; }
        printf("}\n");
        return 0;
    }
    if (!only || !strcmp(only, "itx")) gen_itx(dir);
    if (!only || !strcmp(only, "mc"))  gen_mc(dir);
    if (!only || !strcmp(only, "mcx")) gen_mcx(dir);
    if (!only || !strcmp(only, "mca")) gen_mca(dir);
    if (!only || !strcmp(only, "lmcs")) gen_lmcs(dir);
    if (!only || !strcmp(only, "gpm")) gen_gpm(dir);
    if (!only || !strcmp(only, "dbf")) gen_dbf(dir);
    if (!only || !strcmp(only, "sao")) gen_sao(dir);
    if (!only || !strcmp(only, "alf")) gen_alf(dir);
    if ((!only || !strcmp(only, "intra")) && !g_shim) gen_intra(dir);
    if (!only || !strcmp(only, "intra_ctu")) gen_intra_ctu(dir);
    if (!only || !strcmp(only, "isp")) gen_isp(dir);
    return 0;
}
