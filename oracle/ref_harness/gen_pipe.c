/* gen_pipe.c -- TEST INFRASTRUCTURE (this container only).
 *
 * A reference-driven CHAINED mini-stream: the reference's own slice decoder (libovvc/slicedec.c: slicedec_init_slice_tools ->
 * slicedec_decode_rect_entry -> decode_ctu_line / decode_ctu_last_line -> decode_ctu / decode_truncated_ctu -> coding_quadtree /
 * dual_tree -> coding_unit / prediction_unit / transform_unit -> the rcn slots, with the reference's own CTU scratch, intra line,
 * SAO / ALF line buffers, dbf_store_info / dbf_load_info, store_inter_maps, TMVP planes) decodes a few 416x240 pictures -- 4 x 2 CTUs,
 * last column 32 wide, last row 112 high: SURVEY.md 8 "C1" geometry -- each from the pictures decoded before it.
 *
 * What the harness supplies instead of the parts of the decoder that cannot be built here (ovdec.c / rcn.c / ovmem.c / ovutils.c
 * include the autoconf-generated ovconfig.h; no .266 stream exists on disk):
 *   - parameter sets as filled structs (OVSPS / OVPPS / OVPH / OVSH / OVAPS with the tool set of the JVET CTC random-access
 *     configuration) handed to the reference's own decinit_update_params();
 *   - pictures, reference lists and collocated-picture info as filled OVPicture structs (what dpb.c derives from the RPL syntax);
 *   - every buffer the decoder would take from ov_malloc (line buffers, filter buffers, motion planes), calloc'ed with the sizes
 *     the reference's init functions use (cited below);
 *   - the SLICE DATA: seeded pseudo-random bytes.  The reference's CABAC engine turns uniformly random bits into syntax elements
 *     distributed by its own context models, so the parse is a legal-syntax random walk through the reference's real caller code:
 *     partitions, modes, motion vectors (merge / MMVD / AMVP / affine / SbTMVP / GPM / CIIP / SMVD / BCW / AMVR), residuals
 *     (dependent quantisation, MTS, LFNST, SBT, ISP, transform skip, BDPCM, JCCR), SAO / ALF / CC-ALF CTU parameters.
 *   - rcn_init_functions(): rcn.c is not built, the table is filled by the same per-file initialisers in the same order
 *     (ref_common.h: ref_fill_table, as for every other fixture); in "shim" mode followed by rcn_init_functions_hip() --
 *     exactly the binding INTEGRATION.md section 1 adds to rcn.c.  slicedec.c is compiled as part of this file (it is included
 *     from where it lies, so that its static entry points can be called); the one symbol it takes from rcn.c is redirected here.
 *
 * Reference mode  -> tests/golden/pipe.ovg      : every picture's final frame (after deblocking, SAO, ALF) and the vectors every
 *                                                 rcn_dmvr_mv_refine call returned.
 * Shim mode       -> tests/golden/shim_pipe.ovg : what the INSTALLED slots recorded for the same pictures (one command stream per
 *                                                 picture + its picture-level parameters).  tests/ decode picture k from the
 *                                                 pictures THEY decoded before and must end with the reference's bytes.
 * Device mode     -> tests/golden/shim_pipe_dev.ovg : the shim NOT in record-only mode but with its device half live -- begin_picture,
 *                                                 dpb_get, ref_slot -> ovhip_frame_ref, dmvr_rows_step, flush_picture -- over a device
 *                                                 DPB on a test memory back-end (ovhip_dpb_create_ex) whose frames are dry (nothing
 *                                                 is launched; include/ovvc_hip.h, ovhip_frame_set_trace).  The fixture is the
 *                                                 interleaved log of the decoder's row-end events and the frame-level calls the shim
 *                                                 made under them; what its recorder held must equal the shim mode's stream.
 * Live mode       -> no fixture, a JSON line    : the shim as a decoder would run it -- its own device DPB on the real GPU
 *                                                 (OVVC_HIP_DEVICES), ovhip_frame_submit launching, OVHIP_OUT_PLANES into the OVFrame --
 *                                                 on N frame threads (one OVSliceDec + OVCTUDec each, pictures taken in decoding order
 *                                                 as ovdec_select_subdec hands NAL units to free sub-decoders, ovdec.c:188-248), the
 *                                                 collocated motion field read under the reference's row synchronisation
 *                                                 (drv_mvp.c:281-294; dpb.c:1242-1323).  NOTHING is fed back from the reference pass: the DMVR
 *                                                 slot returns what the shim returns and the TMVP planes hold what the DEVICE delivered
 *                                                 through ovhip_frame_dmvr_rows_collect before the row was reported.  Every picture's
 *                                                 OVFrame and both motion planes are compared with the reference pass in this process.
 * The same process runs the reference pass first in all modes.  In the shim and device (dry) modes the shim's DMVR slot returns
 * unrefined vectors and no device refines them, so there the harness hands the caller the vectors the
 * reference pass produced for the same call and the parse of the later pictures (TMVP) stays the same; the live mode does not.
 */
#define rcn_init_functions gp_rcn_init_functions
#include "ref_common.h"
#include "ovvc_hip.h"
#include "shim_stream.h"
#include "nvcl.h"
#include "nvcl_structures.h"
#include "decinit.h"
#include "ovdec_internal.h"
#include <pthread.h>
#include <time.h>
#include <signal.h>
#include <execinfo.h>
#include <unistd.h>

#include "slicedec.c"        /* /root/reference/libovvc/slicedec.c, compiled where it lies (-I$(R)) */

/* private to rcn_lmcs.c:75-81 (the harness only needs its size: rcn_init_lmcs would take it from ov_malloc) */
struct LMCSLUTs { OVSample fwd_lut[1024]; OVSample bwd_lut[1024]; OVSample wnd_bnd[17]; };

/* ------------------------------------------------------------------------------------------------ the table */
typedef uint8_t (*dmvr_fn)(OVCTUDec *const, struct OVBuffInfo, uint8_t, uint8_t, uint8_t, uint8_t, OVMV *, OVMV *, uint8_t, uint8_t, uint8_t);
static dmvr_fn g_dmvr_inner;
static gbuf g_dmvr_log = { .type = T_I32 };         /* per call: x, y (picture, luma), log2 w, log2 h, mv0 in, mv1 in, mv0 out, mv1 out */
static __thread size_t g_dmvr_pos;                  /* shim pass: next entry of the reference pass's log (per frame thread: set to the picture's first call) */
static int g_pass_shim;                             /* 0 reference slots, 1 installed slots record-only, 2 device half on dry frames, 3 live on the GPU */
static int g_threads;                               /* "threads N[,M,...]": frame threads of the device / live pass (0: the one-thread loop of the fixtures); a list = one pass each */
static int g_thread_list[16], g_n_thread_list;
static int g_tile_cols = 1, g_tile_rows = 1;         /* "tiles C R": C x R rect entries per picture, decoded one after the other on ONE OVCTUDec (slicedec.c:649-653) */
static int g_no_isp;                                /* "noisp": sps_isp_enabled_flag = 0 (long 4K streams: some 64x8 CU is split into 64x2 partitions in nearly every one) */
static int g_allow_64x2;                            /* "allow64x2" (live): a stream that holds 64x2 ISP partitions is decoded all the same -- the back-end follows H.266 there, the reference pass
                                                     * whatever its stack held: frames that differ are counted and reported, not an error (no reference result exists for them) */
static int g_isp_64x2;                              /* the reference's result for 64x2 ISP partitions is undefined (gen_golden.c, gen_isp) */

/* ---- device mode: the decoder's events the shim hangs its frame-level calls on, interleaved with those calls ---- */
enum { GP_EV_ATTACH = 100, GP_EV_SAO_FIRST, GP_EV_ALF_LINE, GP_EV_HOOK_END, GP_EV_DMVR_SLOT };
static gbuf g_events = { .type = T_U8 };
static void
gp_event(uint32_t op, int64_t a, int64_t b)
{
    ovhip_frame_event ev;
    memset(&ev, 0, sizeof(ev));
    ev.op = op; ev.frame = -1; ev.a = a; ev.b = b;
    gbuf_push(&g_events, &ev, sizeof(ev));
}
static void gp_trace_sink(void *user, const ovhip_frame_event *ev) { (void)user; gbuf_push(&g_events, ev, sizeof(*ev)); }

static uint8_t
gp_dmvr(OVCTUDec *const c, struct OVBuffInfo dst, uint8_t x0, uint8_t y0, uint8_t l2w, uint8_t l2h, OVMV *mv0, OVMV *mv1,
        uint8_t ref_idx0, uint8_t ref_idx1, uint8_t apply_bdof)
{
    const int32_t in[8] = { (c->ctb_x << 7) + x0, (c->ctb_y << 7) + y0, l2w, l2h, mv0->x, mv0->y, mv1->x, mv1->y };
    const uint8_t r = __atomic_load_n(&g_dmvr_inner, __ATOMIC_RELAXED)(c, dst, x0, y0, l2w, l2h, mv0, mv1, ref_idx0, ref_idx1, apply_bdof);
    if (g_pass_shim == 2 && !g_threads) gp_event(GP_EV_DMVR_SLOT, in[0], in[1]);
    if (g_pass_shim == 3) return r;                 /* live: what the installed slot returned; the device delivers the refined vectors */
    if (!g_pass_shim) {
        int32_t rec[12];
        memcpy(rec, in, sizeof(in));
        rec[8] = mv0->x; rec[9] = mv0->y; rec[10] = mv1->x; rec[11] = mv1->y;
        gbuf_push(&g_dmvr_log, rec, 12);
        return r;
    }
    const int32_t *want = (const int32_t *)g_dmvr_log.data + 12 * g_dmvr_pos;
    if (12 * (g_dmvr_pos + 1) > g_dmvr_log.n || memcmp(want, in, sizeof(in))) {
        fprintf(stderr, "gen_pipe: shim pass: DMVR call %zu is not the reference pass's call (the parse diverged)\n", g_dmvr_pos);
        exit(1);
    }
    mv0->x = want[8]; mv0->y = want[9]; mv1->x = want[10]; mv1->y = want[11];
    ++g_dmvr_pos;
    return r;
}

typedef void (*isp_fn)(OVCTUDec *const, unsigned int, unsigned int, unsigned int, unsigned int, uint8_t, const struct ISPTUInfo *const);
static isp_fn g_isp_h_inner;
static void
gp_isp_h(OVCTUDec *const c, unsigned int x0, unsigned int y0, unsigned int l2w, unsigned int l2h, uint8_t mode, const struct ISPTUInfo *const tu)
{
    if (l2w == 6 && l2h == 3) __atomic_fetch_add(&g_isp_64x2, 1, __ATOMIC_RELAXED);
    __atomic_load_n(&g_isp_h_inner, __ATOMIC_RELAXED)(c, x0, y0, l2w, l2h, mode, tu);
}

static void (*g_attach_inner)(struct OVRCNCtx *const, const OVFrame *const, const struct RectEntryInfo *const, uint8_t);
static void (*g_sao_first_inner)(OVCTUDec *const, const struct RectEntryInfo *const, uint16_t);
static void (*g_alf_line_inner)(OVCTUDec *const, const struct RectEntryInfo *const, uint16_t);
static void gp_attach(struct OVRCNCtx *const r, const OVFrame *const f, const struct RectEntryInfo *const e, uint8_t l2)
{ gp_event(GP_EV_ATTACH, e->nb_ctu_w, e->nb_ctu_h); g_attach_inner(r, f, e, l2); gp_event(GP_EV_HOOK_END, GP_EV_ATTACH, 0); }
static void gp_sao_first(OVCTUDec *const c, const struct RectEntryInfo *const e, uint16_t y)
{ gp_event(GP_EV_SAO_FIRST, y, e->nb_ctu_h); g_sao_first_inner(c, e, y); gp_event(GP_EV_HOOK_END, GP_EV_SAO_FIRST, y); }
static void gp_alf_line(OVCTUDec *const c, const struct RectEntryInfo *const e, uint16_t y)
{ gp_event(GP_EV_ALF_LINE, y, e->nb_ctu_h); g_alf_line_inner(c, e, y); gp_event(GP_EV_HOOK_END, GP_EV_ALF_LINE, y); }

static void gp_t_attach(struct OVRCNCtx *const r, const OVFrame *const f, const struct RectEntryInfo *const e, uint8_t l2);
static void gp_t_sao_first(OVCTUDec *const c, const struct RectEntryInfo *const e, uint16_t y);
static void gp_t_alf_line(OVCTUDec *const c, const struct RectEntryInfo *const e, uint16_t y);

/* test memory back-end of the device DPB: a "picture" is a number */
static int fm_next = 0x1000;
static int fm_alloc(void *u, int dev, int32_t w, int32_t h, ovhip_pic *pic) { (void)u; (void)dev; memset(pic, 0, sizeof(*pic)); pic->y = (uint16_t *)(uintptr_t)__atomic_add_fetch(&fm_next, 0x100, __ATOMIC_RELAXED); pic->w = w; pic->h = h; pic->stride_y = w; pic->stride_c = w / 2; return 0; }
static void fm_free(void *u, int dev, ovhip_pic *pic) { (void)u; (void)dev; (void)pic; }
static int fm_copy_start(void *u, int dd, const ovhip_pic *dst, int sd, const ovhip_pic *src, void **ev) { (void)u; (void)dd; (void)dst; (void)sd; (void)src; *ev = NULL; return 0; }
static int fm_copy_wait(void *u, int dev, void *ev) { (void)u; (void)dev; (void)ev; return 0; }

typedef void (*gpm_fn)(OVCTUDec *const, struct VVCGPM *, int, int, int, int);
static gpm_fn g_gpm_inner;
static void
gp_gpm(OVCTUDec *const c, struct VVCGPM *g, int x0, int y0, int l2w, int l2h)
{
    if (getenv("GP_TRACE"))
        fprintf(stderr, "    gpm %s ctb %d,%d at %d,%d %dx%d split %d dir %d/%d mv0 %d,%d ref %d bcw %d mv1 %d,%d ref %d\n", g_pass_shim ? "shim" : "ref", c->ctb_x, c->ctb_y,
                x0, y0, 1 << l2w, 1 << l2h, g->split_dir, g->inter_dir0, g->inter_dir1, g->mv0.x, g->mv0.y, g->mv0.ref_idx, g->mv0.bcw_idx_plus1, g->mv1.x, g->mv1.y, g->mv1.ref_idx);
    g_gpm_inner(c, g, x0, y0, l2w, l2h);
}

/* rcn.c:147-180 for 10-bit, + the MI355X override block when the shim pass runs */
void
gp_rcn_init_functions(struct RCNFunctions *f, uint8_t ict_type, uint8_t lm_chroma_enabled, uint8_t vcolloc, uint8_t lmcs_flag, uint8_t bitdepth)
{
    if (bitdepth != 10 || !lm_chroma_enabled || vcolloc) { fprintf(stderr, "gen_pipe: table variant not restated\n"); exit(1); }
    ref_fill_table(f, ict_type, lmcs_flag);
    if (g_pass_shim) rcn_init_functions_hip(f, ict_type, lm_chroma_enabled, vcolloc, lmcs_flag, bitdepth);
    /* (every frame thread fills its own table with the same slots: the "inner" pointers are the same values, stored relaxed) */
#define GP_WRAP(inner, slot, wrapper) do { __atomic_store_n(&(inner), (slot), __ATOMIC_RELAXED); (slot) = (wrapper); } while (0)
    GP_WRAP(g_dmvr_inner, f->rcn_dmvr_mv_refine, &gp_dmvr);
    if (getenv("GP_TRACE")) GP_WRAP(g_gpm_inner, f->rcn_gpm_b, &gp_gpm);
    if (g_pass_shim >= 2 && g_threads) {
        GP_WRAP(g_attach_inner, f->rcn_attach_frame_buff, &gp_t_attach);
        GP_WRAP(g_sao_first_inner, f->sao.rcn_sao_first_pix_rows, &gp_t_sao_first);
        GP_WRAP(g_alf_line_inner, f->alf.rcn_alf_filter_line, &gp_t_alf_line);
    } else if (g_pass_shim == 2) {
        GP_WRAP(g_attach_inner, f->rcn_attach_frame_buff, &gp_attach);
        GP_WRAP(g_sao_first_inner, f->sao.rcn_sao_first_pix_rows, &gp_sao_first);
        GP_WRAP(g_alf_line_inner, f->alf.rcn_alf_filter_line, &gp_alf_line);
    }
    { isp_fn cur = (isp_fn)f->tmp.recon_isp_subtree_h; __atomic_store_n(&g_isp_h_inner, cur, __ATOMIC_RELAXED); f->tmp.recon_isp_subtree_h = (void *)&gp_isp_h; }
#undef GP_WRAP
}

/* ------------------------------------------------------------------------------------------------ parameter sets */
struct gp_seq {
    int w, h, nb_ctb_w, nb_ctb_h;
    OVSPS sps; OVPPS pps; OVPH ph; OVSH sh;
    OVAPS aps_alf[4], aps_lmcs;
    OVNVCLCtx nvcl;
    OVPS ps;
};

static void
rand_alf_aps(OVALFData *a)          /* as gen_golden.c, gen_alf */
{
    memset(a, 0, sizeof(*a));
    int nf = rnd_range(1, 25);
    a->alf_luma_num_filters_signalled_minus1 = nf - 1;
    a->alf_luma_clip_flag = rnd_range(0, 1);
    for (int c = 0; c < 25; ++c) a->alf_luma_coeff_delta_idx[c] = rnd_range(0, nf - 1);
    for (int f = 0; f < 25; ++f) for (int k = 0; k < 12; ++k) {
        a->alf_luma_coeff[f][k] = (int16_t)(rnd_range(0, 7) == 0 ? rnd_range(-60, 60) : rnd_range(-12, 12));
        a->alf_luma_clip_idx[f][k] = rnd_range(0, 3);
    }
    a->alf_chroma_clip_flag = rnd_range(0, 1);
    a->alf_chroma_num_alt_filters_minus1 = rnd_range(0, 7);
    for (int f = 0; f < 8; ++f) for (int k = 0; k < 6; ++k) {
        a->alf_chroma_coeff[f][k] = (int16_t)(rnd_range(0, 7) == 0 ? rnd_range(-60, 60) : rnd_range(-12, 12));
        a->alf_chroma_clip_idx[f][k] = rnd_range(0, 3);
    }
    a->alf_cc_cb_filters_signalled_minus1 = 3; a->alf_cc_cr_filters_signalled_minus1 = 3;
    for (int c = 0; c < 2; ++c) for (int f = 0; f < 4; ++f) for (int k = 0; k < 7; ++k) {
        int m = rnd_range(0, 5);
        int v = m == 0 ? 0 : 1 << (m - 1);
        a->alf_cc_mapped_coeff[c][f][k] = (int16_t)(rnd_range(0, 1) ? -v : v);
    }
}

/* The sequence-level tool set: JVET CTC random access (VTM encoder_randomaccess_vtm.cfg), 10-bit 4:2:0, CTU 128. */
static void
seq_init(struct gp_seq *s, int w, int h, int variant)
{
    memset(s, 0, sizeof(*s));
    s->w = w; s->h = h; s->nb_ctb_w = (w + 127) >> 7; s->nb_ctb_h = (h + 127) >> 7;
    OVSPS *sps = &s->sps;
    sps->sps_chroma_format_idc = 1;
    sps->sps_log2_ctu_size_minus5 = 2;
    sps->sps_pic_width_max_in_luma_samples = w; sps->sps_pic_height_max_in_luma_samples = h;
    sps->sps_bitdepth_minus8 = 2;
    sps->sps_log2_min_luma_coding_block_size_minus2 = 0;
    sps->sps_log2_diff_min_qt_min_cb_intra_slice_luma = 1;  sps->sps_max_mtt_hierarchy_depth_intra_slice_luma = 3;
    sps->sps_log2_diff_max_bt_min_qt_intra_slice_luma = 2;  sps->sps_log2_diff_max_tt_min_qt_intra_slice_luma = 2;
    sps->sps_qtbtt_dual_tree_intra_flag = variant != 1;
    sps->sps_log2_diff_min_qt_min_cb_intra_slice_chroma = 1; sps->sps_max_mtt_hierarchy_depth_intra_slice_chroma = 3;
    sps->sps_log2_diff_max_bt_min_qt_intra_slice_chroma = 3; sps->sps_log2_diff_max_tt_min_qt_intra_slice_chroma = 2;
    sps->sps_log2_diff_min_qt_min_cb_inter_slice = 1;        sps->sps_max_mtt_hierarchy_depth_inter_slice = 3;
    sps->sps_log2_diff_max_bt_min_qt_inter_slice = 4;        sps->sps_log2_diff_max_tt_min_qt_inter_slice = 3;
    sps->sps_max_luma_transform_size_64_flag = 1;
    sps->sps_transform_skip_enabled_flag = 1; sps->sps_log2_transform_skip_max_size_minus2 = 3; sps->sps_bdpcm_enabled_flag = 1;
    sps->sps_mts_enabled_flag = 1; sps->sps_explicit_mts_intra_enabled_flag = variant != 1; sps->sps_explicit_mts_inter_enabled_flag = variant == 1;
    sps->sps_lfnst_enabled_flag = 1;
    sps->sps_joint_cbcr_enabled_flag = 1;
    sps->sps_same_qp_table_for_chroma_flag = 1;
    sps->sps_qp_table_start_minus26[0] = -9; sps->sps_num_points_in_qp_table_minus1[0] = 2;     /* 17 22 34 42 -> 17 23 35 39 */
    sps->sps_delta_qp_in_val_minus1[0][0] = 4; sps->sps_delta_qp_in_val_minus1[0][1] = 11; sps->sps_delta_qp_in_val_minus1[0][2] = 7;
    sps->sps_delta_qp_diff_val[0][0] = 4 ^ 6;  sps->sps_delta_qp_diff_val[0][1] = 11 ^ 12;    sps->sps_delta_qp_diff_val[0][2] = 7 ^ 4;
    sps->sps_sao_enabled_flag = 1; sps->sps_alf_enabled_flag = 1; sps->sps_ccalf_enabled_flag = 1; sps->sps_lmcs_enabled_flag = 1;
    sps->sps_temporal_mvp_enabled_flag = 1; sps->sps_sbtmvp_enabled_flag = 1; sps->sps_amvr_enabled_flag = 1;
    sps->sps_bdof_enabled_flag = 1; sps->sps_smvd_enabled_flag = 1; sps->sps_dmvr_enabled_flag = 1;
    sps->sps_mmvd_enabled_flag = 1; sps->sps_mmvd_fullpel_only_enabled_flag = 0;
    sps->sps_six_minus_max_num_merge_cand = 0;
    sps->sps_sbt_enabled_flag = 1;
    sps->sps_affine_enabled_flag = 1; sps->sps_five_minus_max_num_subblock_merge_cand = 0; sps->sps_6param_affine_enabled_flag = 1;
    sps->sps_affine_amvr_enabled_flag = 1; sps->sps_affine_prof_enabled_flag = 1;
    sps->sps_bcw_enabled_flag = 1; sps->sps_ciip_enabled_flag = 1;
    sps->sps_gpm_enabled_flag = 1; sps->sps_max_num_merge_cand_minus_max_num_gpm_cand = 1;
    rcn_init_gpm_params();                         /* what the SPS reader does when it meets the flag (nvcl_nal_sps.c:580-583) */
    sps->sps_isp_enabled_flag = !g_no_isp; sps->sps_mrl_enabled_flag = 1; sps->sps_mip_enabled_flag = 1; sps->sps_cclm_enabled_flag = 1;
    sps->sps_chroma_horizontal_collocated_flag = 1; sps->sps_chroma_vertical_collocated_flag = 0;
    sps->sps_ibc_enabled_flag = 0;
    sps->sps_dep_quant_enabled_flag = 1; sps->sps_sign_data_hiding_enabled_flag = 1;

    OVPPS *pps = &s->pps;
    pps->pps_pic_width_in_luma_samples = w; pps->pps_pic_height_in_luma_samples = h;
    pps->pps_log2_ctu_size_minus5 = 2;
    pps->pps_no_pic_partition_flag = 1;
    if (g_tile_cols * g_tile_rows > 1) {
        /* tiles: uniform columns / rows, the last one takes what is left (what pps_read derives from the explicit widths;
         * nvcl_nal_pps.c) -- dec_init.c:369-404 reads exactly these fields.  One rect entry per tile (slicedec.c:636-657). */
        const int nw = (w + 127) >> 7, nh = (h + 127) >> 7;
        if (g_tile_cols > nw || g_tile_rows > nh) { fprintf(stderr, "gen_pipe: more tiles than CTUs\n"); exit(2); }
        pps->pps_no_pic_partition_flag = 0;
        pps->pps_num_tile_columns_minus1 = g_tile_cols - 1; pps->pps_num_tile_rows_minus1 = g_tile_rows - 1;
        const int cw = (nw + g_tile_cols - 1) / g_tile_cols, ch = (nh + g_tile_rows - 1) / g_tile_rows;
        for (int i = 0, left = nw; i < g_tile_cols; ++i) { const int n = i == g_tile_cols - 1 ? left : (cw < left - (g_tile_cols - 1 - i) ? cw : left - (g_tile_cols - 1 - i)); pps->pps_tile_column_width_minus1[i] = n - 1; left -= n; }
        for (int i = 0, left = nh; i < g_tile_rows; ++i) { const int n = i == g_tile_rows - 1 ? left : (ch < left - (g_tile_rows - 1 - i) ? ch : left - (g_tile_rows - 1 - i)); pps->pps_tile_row_height_minus1[i] = n - 1; left -= n; }
        pps->pps_loop_filter_across_tiles_enabled_flag = 0;        /* (never read by the reference: a rect entry is filtered on its own) */
    }
    pps->pps_init_qp_minus26 = 0;
    pps->pps_cu_qp_delta_enabled_flag = variant == 1;
    pps->pps_cb_qp_offset = variant == 1 ? 1 : 0; pps->pps_cr_qp_offset = variant == 1 ? -1 : 0;
    pps->pps_joint_cbcr_qp_offset_value = 0;

    for (int i = 0; i < 4; ++i) { s->aps_alf[i].aps_adaptation_parameter_set_id = i; rand_alf_aps(&s->aps_alf[i].aps_alf_data); }
    struct OVLMCSData *ld = &s->aps_lmcs.aps_lmcs_data;
    ld->lmcs_min_bin_idx = 1; ld->lmcs_delta_max_bin_idx = 2;
    for (int i = 0; i < 16; ++i) { ld->lmcs_delta_abs_cw[i] = rnd_range(0, 24); ld->lmcs_delta_sign_cw_flag[i] = rnd_range(0, 1); }
    ld->lmcs_delta_abs_crs = rnd_range(0, 5); ld->lmcs_delta_sign_crs_flag = rnd_range(0, 1);

    s->nvcl.sps_list[0] = sps; s->nvcl.pps_list[0] = pps;
    for (int i = 0; i < 4; ++i) s->nvcl.alf_aps_list[i] = &s->aps_alf[i];
    s->nvcl.lmcs_aps_list[0] = &s->aps_lmcs;
    s->nvcl.ph = &s->ph; s->nvcl.sh = &s->sh;
}

/* picture- and slice-level syntax of one picture (what nvcl_ph_read / nvcl_sh_read would have filled) */
static void
pic_headers(struct gp_seq *s, int slice_type, int qp, int n_ref0, int n_ref1, int tmvp, int col_from_l0, int lmcs)
{
    OVPH *ph = &s->ph; OVSH *sh = &s->sh;
    memset(ph, 0, sizeof(*ph)); memset(sh, 0, sizeof(*sh));
    ph->ph_inter_slice_allowed_flag = slice_type != 2; ph->ph_intra_slice_allowed_flag = 1;
    ph->ph_lmcs_enabled_flag = lmcs; ph->ph_lmcs_aps_id = 0; ph->ph_chroma_residual_scale_flag = lmcs;
    ph->ph_temporal_mvp_enabled_flag = tmvp; ph->ph_collocated_from_l0_flag = col_from_l0; ph->ph_collocated_ref_idx = 0;
    ph->ph_joint_cbcr_sign_flag = rnd_range(0, 1);
    ph->ph_mmvd_fullpel_only_flag = 0; ph->ph_mvd_l1_zero_flag = 0;
    ph->ph_cu_qp_delta_subdiv_intra_slice = 2; ph->ph_cu_qp_delta_subdiv_inter_slice = 2;
    ph->ph_qp_delta = 0;
    sh->sh_slice_type = slice_type;
    sh->sh_qp_delta = qp - 26;
    sh->sh_alf_enabled_flag = 1; sh->sh_num_alf_aps_ids_luma = 2; sh->sh_alf_aps_id_luma[0] = 0; sh->sh_alf_aps_id_luma[1] = 1;
    sh->sh_alf_cb_enabled_flag = 1; sh->sh_alf_cr_enabled_flag = 1; sh->sh_alf_aps_id_chroma = 2;
    sh->sh_alf_cc_cb_enabled_flag = 1; sh->sh_alf_cc_cb_aps_id = 3; sh->sh_alf_cc_cr_enabled_flag = 1; sh->sh_alf_cc_cr_aps_id = 3;
    sh->sh_lmcs_used_flag = lmcs;
    sh->sh_sao_luma_used_flag = 1; sh->sh_sao_chroma_used_flag = 1;
    /* debugging aids (never set when the committed fixtures are made): switch in-loop filter stages off */
    if (getenv("GP_NO_SAO")) sh->sh_sao_luma_used_flag = sh->sh_sao_chroma_used_flag = 0;
    if (getenv("GP_NO_ALF")) sh->sh_alf_enabled_flag = sh->sh_alf_cb_enabled_flag = sh->sh_alf_cr_enabled_flag = sh->sh_alf_cc_cb_enabled_flag = sh->sh_alf_cc_cr_enabled_flag = 0;
    if (getenv("GP_NO_DBF")) sh->sh_deblocking_filter_disabled_flag = 1;
    sh->sh_collocated_from_l0_flag = col_from_l0;
    sh->sh_dep_quant_used_flag = rnd_range(0, 3) != 0; sh->sh_sign_data_hiding_used_flag = !sh->sh_dep_quant_used_flag;
    sh->sh_luma_beta_offset_div2 = rnd_range(-2, 2); sh->sh_luma_tc_offset_div2 = rnd_range(-2, 2);
    sh->hrpl.rpl_h0.rpl_data.num_ref_active_entries = n_ref0;
    sh->hrpl.rpl_h1.rpl_data.num_ref_active_entries = n_ref1;
}

/* ------------------------------------------------------------------------------------------------ pictures */
static OVPicture *
gp_new_picture(const struct gp_seq *s, int poc)
{
    OVPicture *p = ref_new_picture(s->w, s->h, poc);
    /* The planes of ref_new_picture are tight callocs; the decoder's frame pool hands out planes with mapped memory around them.  The
     * reference's DMVR window fetch reads a few samples past a plane's end for blocks at the bottom picture border (seed 1247 at
     * 3840x2160: a fault in rcn_dmvr_mv_refine -- the values are never used, the fixtures match with and without the slack): give
     * every plane 16 rows of zeroed slack on both sides, as any real allocation has. */
    for (int k = 0; k < 3; ++k) {
        const size_t wk = (size_t)(k ? s->w / 2 : s->w), hk = (size_t)(k ? s->h / 2 : s->h);
        free(p->frame->data[k]);
        p->frame->data[k] = (uint8_t *)calloc(wk * (hk + 32), 2) + wk * 16 * 2;
    }
    /* ovdpb_init_decoded_ctus (dpb.c:1272-1295) */
    p->decoded_ctus.mask_h = s->nb_ctb_h; p->decoded_ctus.mask_w = (s->nb_ctb_w >> 6) + 1;
    p->decoded_ctus.mask = calloc(s->nb_ctb_h, sizeof(uint64_t *));
    for (int i = 0; i < s->nb_ctb_h; ++i) p->decoded_ctus.mask[i] = calloc(p->decoded_ctus.mask_w, 8);
    pthread_mutex_init(&p->internal.ref_mtx, NULL); pthread_cond_init(&p->internal.ref_cnd, NULL);
    p->decoded_ctus.ref_mtx = &p->internal.ref_mtx; p->decoded_ctus.ref_cnd = &p->internal.ref_cnd;
    /* mvpool.c:52-82: one OVMV per 8x8 block, one direction word per 4-sample row of every CTU */
    const size_t n_ctb = (size_t)s->nb_ctb_w * s->nb_ctb_h;
    p->mv_plane0.mvs = calloc(n_ctb * 16 * 16, sizeof(OVMV)); p->mv_plane0.dirs = calloc(n_ctb * 32, 8);
    p->mv_plane1.mvs = calloc(n_ctb * 16 * 16, sizeof(OVMV)); p->mv_plane1.dirs = calloc(n_ctb * 32, 8);
    return p;
}

/* reference lists + what init_tmvp_info / tmvp_set_mv_scales (dpb.c:951-1062) derive from them */
static void
gp_set_refs(OVPicture *p, OVPicture **l0, int n0, OVPicture **l1, int n1, int tmvp, int col_from_l0)
{
    memset(p->rpl0, 0, sizeof(p->rpl0)); memset(p->rpl1, 0, sizeof(p->rpl1));
    memset(&p->rpl_info0, 0, sizeof(p->rpl_info0)); memset(&p->rpl_info1, 0, sizeof(p->rpl_info1));
    for (int i = 0; i < n0; ++i) { p->rpl0[i] = l0[i]; p->rpl_info0.ref_info[i].type = ST_REF; p->rpl_info0.ref_info[i].poc = l0[i]->poc; }
    for (int i = 0; i < n1; ++i) { p->rpl1[i] = l1[i]; p->rpl_info1.ref_info[i].type = ST_REF; p->rpl_info1.ref_info[i].poc = l1[i]->poc; }
    p->rpl_info0.nb_refs = p->rpl_info0.nb_active_refs = n0; p->rpl_info1.nb_refs = p->rpl_info1.nb_active_refs = n1;
    for (int i = 0; i < n0; ++i) p->tmvp.dist_ref_0[i] = p->poc - l0[i]->poc;
    for (int i = 0; i < n1; ++i) p->tmvp.dist_ref_1[i] = p->poc - l1[i]->poc;
    p->tmvp.collocated_ref = NULL;
    if (tmvp && (n0 || n1)) {
        const OVPicture *col = col_from_l0 ? l0[0] : l1[0];
        p->tmvp.collocated_ref = col;
        p->tmvp.col_info.ref_idx_rpl0 = col_from_l0 ? 0 : -1; p->tmvp.col_info.ref_idx_rpl1 = col_from_l0 ? -1 : 0;
        for (int i = 0; i < 16; ++i) {
            if (col_from_l0 && p->rpl1[i] == col) p->tmvp.col_info.ref_idx_rpl1 = i;
            if (!col_from_l0 && p->rpl0[i] == col) p->tmvp.col_info.ref_idx_rpl0 = i;
        }
        for (int i = 0; i < col->rpl_info0.nb_refs; ++i) p->tmvp.dist_col_0[i] = col->poc - col->rpl_info0.ref_info[i].poc;
        for (int i = 0; i < col->rpl_info1.nb_refs; ++i) p->tmvp.dist_col_1[i] = col->poc - col->rpl_info1.ref_info[i].poc;
    }
}

/* ------------------------------------------------------------------------------------------------ slice decoder + entry decoder */
static void
gp_alloc_lines(OVSliceDec *sl, const struct gp_seq *s)
{
    /* init_cabac_lines (slicedec.c:357-391): one byte per 4-sample unit of the picture width, per tile row */
    const int nb_pb = (s->nb_ctb_w << 5) * g_tile_rows;
    for (int k = 0; k < 2; ++k) {
        sl->cabac_lines[k].qt_depth_map_x = calloc(nb_pb, 1); sl->cabac_lines[k].log2_cu_w_map_x = calloc(nb_pb, 1); sl->cabac_lines[k].cu_mode_x = calloc(nb_pb, 1);
    }
    /* init_drv_lines (drv_lines.c:772-817): nb_ctb_pic_w + 2 per tile column, per tile row */
    const int nb_ctb = (s->nb_ctb_w + 2 * g_tile_cols) * g_tile_rows, nb_pb2 = nb_ctb << 5, n_in = 32 * nb_ctb + 2;
    struct DRVLines *l = &sl->drv_lines;
    l->inter_lines.mv0 = calloc((size_t)32 * n_in, sizeof(OVMV)); l->inter_lines.mv1 = calloc((size_t)32 * n_in, sizeof(OVMV));
    l->inter_lines.dir0 = calloc(n_in, 4); l->inter_lines.dir1 = calloc(n_in, 4); l->inter_lines.affine = calloc(n_in, 4);
    l->inter_lines.aff_info = calloc(n_in, sizeof(struct AffineInfo));
    struct DBFLines *d = &l->dbf_lines;
    d->qp_x_map = calloc(32 * nb_pb2 + 1, 1); d->qp_x_map_cb = calloc(32 * nb_pb2 + 1, 1); d->qp_x_map_cr = calloc(32 * nb_pb2 + 1, 1);
    d->small_map = calloc(nb_ctb + 1, 8); d->large_map_c = calloc(nb_ctb + 1, 8);
    d->dbf_bs1_hor = calloc(nb_ctb + 1, 8); d->dbf_bs1_hor_cb = calloc(nb_ctb + 1, 8); d->dbf_bs1_hor_cr = calloc(nb_ctb + 1, 8);
    d->dbf_bs2_hor = calloc(nb_ctb + 1, 8); d->dbf_bs2_hor_c = calloc(nb_ctb + 1, 8); d->dbf_affine = calloc(nb_ctb + 1, 8);
    l->ibc_lines.mv = calloc((size_t)32 * n_in, sizeof(IBCMV)); l->ibc_lines.map = calloc(n_in, 4);
    l->intra_luma_x = calloc(nb_pb2, 1);
}

static OVCTUDec *
gp_new_ctudec(const struct gp_seq *s)
{
    OVCTUDec *c = NULL;
    if (posix_memalign((void **)&c, 64, sizeof(*c))) abort();
    memset(c, 0, sizeof(*c));
    c->rcn_ctx.ctudec = c;
    const int n_ctb = s->nb_ctb_w * s->nb_ctb_h;
    /* ctudec_init_in_loop_filters (ctudec.c:104-150) allocates these on first use; rcn_init_lmcs (rcn_lmcs.c:347-349) likewise */
    c->sao_info.sao_params = calloc(n_ctb, sizeof(SAOParamsCtu));
    c->alf_info.ctb_alf_params = calloc(n_ctb, sizeof(ALFParamsCtu));
    c->alf_info.ctb_cc_alf_filter_idx[0] = calloc(n_ctb, 1); c->alf_info.ctb_cc_alf_filter_idx[1] = calloc(n_ctb, 1);
    c->lmcs_info.luts = calloc(1, sizeof(struct LMCSLUTs));
    /* rcn_alloc_filter_buffers / rcn_alloc_intra_line_buff (rcn_ctu.c:512-551, :231-244; slicedec.c:1318-1323) */
    struct OVFilterBuffers *fb = &c->rcn_ctx.filter_buffers;
    const int margin = 3;
    fb->margin = margin;
    for (int comp = 0; comp < 3; ++comp) {
        const int ratio = comp ? 2 : 1;
        fb->filter_region_w[comp] = 128 / ratio; fb->filter_region_h[comp] = 128 / ratio;
        fb->filter_region_stride[comp] = 128 / ratio + 2 * margin;
        fb->filter_region_offset[comp] = margin * fb->filter_region_stride[comp] + margin;
        fb->filter_region[comp] = calloc(fb->filter_region_stride[comp] * (fb->filter_region_h[comp] + 2 * margin + 1), sizeof(OVSample));
        fb->saved_cols[comp] = calloc(fb->filter_region_h[comp] * margin, sizeof(OVSample));
        fb->saved_rows_stride[comp] = s->nb_ctb_w * 128 / ratio;
        fb->saved_rows_sao[comp] = calloc(margin * fb->saved_rows_stride[comp], sizeof(OVSample));
        fb->saved_rows_alf[comp] = calloc(margin * fb->saved_rows_stride[comp], sizeof(OVSample));
    }
    struct OVBuffInfo *il = &c->rcn_ctx.intra_line_buff;
    il->stride = (s->nb_ctb_w + 2) << 7; il->stride_c = il->stride >> 1;
    il->y = calloc(il->stride, sizeof(OVSample)); il->cb = calloc(il->stride_c, sizeof(OVSample)); il->cr = calloc(il->stride_c, sizeof(OVSample));
    c->prev_nb_ctu_w_rect_entry = s->nb_ctb_w;
    return c;
}

/* ------------------------------------------------------------------------------------------------ the stream */
#define GP_MAX_PIC 257                                /* I + 32 GOPs of 8 */
struct gp_pic_desc { int poc, slice_type, qp, l0[2], n0, l1[2], n1, tmvp, col_from_l0, lmcs; };

struct gp_out {
    gbuf frames;                               /* uint16: every picture's y, cb, cr */
    gbuf info;                                 /* int32 per picture: poc, slice type, qp, n0, l0[2], n1, l1[2], tmvp, first / end DMVR call */
    struct shim_stream S, S2;                  /* S2: the device pass's recorder contents (must equal S) */
    gbuf sao, alf, tab[5], luts, offs, refmap, pflags;
};

/* What the reference pass leaves behind for the passes that run on frame threads: the picture's slice data, its picture / slice
 * header (own copies: pic_headers draws from the seeded generator, which only the main thread may touch), its first DMVR call in
 * the log, and the decoded picture itself (frame + both collocated motion planes) as the thing to compare with. */
struct gp_kept {
    uint8_t *payload;
    OVPH ph; OVSH sh; OVNVCLCtx nvcl;
    size_t dm0, dm1;
    OVPicture *ref_pic;                            /* reference pass */
    OVPicture *pic;                                /* threaded pass */
};
static struct gp_kept *g_kept;
static uint8_t *g_payload;
static size_t g_payload_bytes = 1 << 20;           /* slice data per picture: 1 MiB covers 1024x1024 many times over; "size" scales it */
#define GP_PAYLOAD g_payload_bytes
static int g_null_shim;
static int g_time_only;                            /* "time" mode: nothing is kept */
static double g_decode_seconds_pass[3];            /* ... per pass: reference slots / installed shim slots (record-only) */
static double g_decode_seconds;                    /* "time" mode: wall time inside slicedec_decode_rect_entry */
static inline double gp_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }

static void
run_stream(struct gp_seq *s, const struct gp_pic_desc *desc, int n_pic, uint32_t seed, struct gp_out *out)
{
    OVPicture *pics[GP_MAX_PIC];
    OVSliceDec sl;
    memset(&sl, 0, sizeof(sl));
    gp_alloc_lines(&sl, s);
    OVCTUDec *c = gp_new_ctudec(s);
    ovhip_dpb *dpb = NULL;
    if (g_pass_shim == 1) {
        /* the entry exists once the table has been installed; bind a recorder to it (record-only: no device in this container) */
        gp_rcn_init_functions(&c->rcn_funcs, 0, 1, 0, 1, 10);
        c->part_ctx = &g_part;
        ovhip_recorder *r = ovhip_rec_create(s->w, s->h);
        if (!g_null_shim && (!r || ovhip_shim_bind_recorder(c, r, s->w, s->h))) { fprintf(stderr, "gen_pipe: shim bind failed\n"); exit(1); }
    } else if (g_pass_shim == 2) {
        /* the device half live: the shim's own begin_picture / flush_picture over a DPB without a device (dry frames) */
        ovhip_dpb_ops ops;
        memset(&ops, 0, sizeof(ops));
        ops.pic_alloc = fm_alloc; ops.pic_free = fm_free; ops.copy_start = fm_copy_start; ops.copy_wait = fm_copy_wait;
        if (ovhip_dpb_create_ex(&dpb, 1, &ops)) { fprintf(stderr, "gen_pipe: ovhip_dpb_create_ex failed\n"); exit(1); }
        ovhip_shim_set_dpb(dpb);
        ovhip_frame_set_trace(gp_trace_sink, NULL);
    }
    for (int k = 0; k < n_pic; ++k) {
        const struct gp_pic_desc *d = &desc[k];
        g_seed = seed + 977 * k;
        if (g_threads && !g_pass_shim && posix_memalign((void **)&g_payload, 64, GP_PAYLOAD)) abort();      /* kept: the frame threads parse the same bytes */
        /* slice data: seeded bytes; the first byte keeps the arithmetic decoder's start condition (vcl_cabac.c:960) */
        for (size_t i = 0; i < GP_PAYLOAD; ++i) g_payload[i] = (uint8_t)(rnd32() >> 7);
        g_payload[0] &= 0x7f;
        pics[k] = gp_new_picture(s, d->poc);
        OVPicture *l0[2] = { d->n0 > 0 ? pics[d->l0[0]] : NULL, d->n0 > 1 ? pics[d->l0[1]] : NULL };
        OVPicture *l1[2] = { d->n1 > 0 ? pics[d->l1[0]] : NULL, d->n1 > 1 ? pics[d->l1[1]] : NULL };
        gp_set_refs(pics[k], l0, d->n0, l1, d->n1, d->tmvp, d->col_from_l0);
        pic_headers(s, d->slice_type, d->qp, d->n0, d->n1, d->tmvp, d->col_from_l0, d->lmcs);
        if (g_threads && !g_pass_shim) {
            struct gp_kept *kp = &g_kept[k];
            kp->payload = g_payload; kp->ph = s->ph; kp->sh = s->sh; kp->nvcl = s->nvcl; kp->nvcl.ph = &kp->ph; kp->nvcl.sh = &kp->sh;
            kp->ref_pic = pics[k];
        }
        s->ps.ph = NULL; s->ps.sh = NULL;               /* new headers in the same structs */
        if (decinit_update_params(&s->ps, &s->nvcl) < 0) { fprintf(stderr, "gen_pipe: decinit_update_params failed\n"); exit(1); }
        /* one rect entry per tile, each with its own stretch of the slice data (decinit_set_entry_points, dec_init.c:320-366) */
        const int n_entries = g_tile_cols * g_tile_rows;
        const size_t per_entry = (GP_PAYLOAD / n_entries) & ~(size_t)63;
        for (int i = 0; i < n_entries; ++i) { s->ps.sh_info.rbsp_entry[i] = g_payload + i * per_entry; g_payload[i * per_entry] &= 0x7f; }
        s->ps.sh_info.rbsp_entry[n_entries] = g_payload + n_entries * per_entry;
        sl.pic = pics[k]; sl.active_params = &s->ps; sl.slice_type = d->slice_type;
        slicedec_init_lines(&sl, &s->ps);
        const size_t dm0 = g_pass_shim ? g_dmvr_pos : g_dmvr_log.n / 12;
        const double t_dec = gp_now();
        for (int i = 0; i < n_entries; ++i) {
            /* slicedec_decode_rect_entries without entry threads (slicedec.c:649-653) */
            slicedec_update_entry_decoder(&sl, c);
            slicedec_decode_rect_entry(&sl, c, &s->ps, i);
        }
        g_decode_seconds += gp_now() - t_dec;
        g_decode_seconds_pass[g_pass_shim] += gp_now() - t_dec;
        ovdpb_report_decoded_frame(pics[k]);
        if ((size_t)(c->cabac_ctx ? 0 : 0)) {}
        const size_t dm1 = g_pass_shim ? g_dmvr_pos : g_dmvr_log.n / 12;
        if (g_threads && !g_pass_shim) { g_kept[k].dm0 = dm0; g_kept[k].dm1 = dm1; }
        if (!g_threads || n_pic <= 9) fprintf(stderr, "  picture %d: POC %d %s qp %d, %zu DMVR calls\n", k, d->poc, d->slice_type == 2 ? "I" : d->slice_type == 1 ? "P" : "B", d->qp, dm1 - dm0);
        if (g_time_only) { if (g_pass_shim && !g_null_shim) { ovhip_shim_flush_pending(c); if (ovhip_shim_last_error(c)) { fprintf(stderr, "gen_pipe: picture %d: the shim latched %d\n", k, ovhip_shim_last_error(c)); exit(1); } } continue; }
        if (!g_pass_shim && g_threads) continue;            /* the picture itself is kept (g_kept[k].ref_pic) */
        if (!g_pass_shim) {
            const OVFrame *f = pics[k]->frame;
            gbuf_push(&out->frames, f->data[0], (size_t)s->w * s->h);
            gbuf_push(&out->frames, f->data[1], (size_t)s->w * s->h / 4);
            gbuf_push(&out->frames, f->data[2], (size_t)s->w * s->h / 4);
            int32_t info[14] = { d->poc, d->slice_type, d->qp, d->n0, d->l0[0], d->l0[1], d->n1, d->l1[0], d->l1[1], d->tmvp, (int32_t)dm0, (int32_t)dm1, d->lmcs, 0 };
            gbuf_push(&out->info, info, 14);
            continue;
        }
        /* shim pass: what the installed slots recorded for this picture */
        if (ovhip_shim_last_error(c)) { fprintf(stderr, "gen_pipe: picture %d: the shim latched %d\n", k, ovhip_shim_last_error(c)); exit(1); }
        size_t nc = 0, nt = 0;
        const int n_ctb = s->nb_ctb_w * s->nb_ctb_h;
        if (g_pass_shim == 2) goto stream_only;
        /* a picture without SAO / ALF: all-off parameters (the flush would get no table at all) */
        const ovhip_sao_ctu *sp = ovhip_shim_sao_params(c, &nc);
        if (sp && nc != (size_t)n_ctb) { fprintf(stderr, "gen_pipe: picture %d: SAO parameters of %zu CTUs\n", k, nc); exit(1); }
        void *zero = calloc(n_ctb, sizeof(*sp) > sizeof(ovhip_alf_ctu) ? sizeof(*sp) : sizeof(ovhip_alf_ctu));
        gbuf_push(&out->sao, sp ? (const void *)sp : zero, n_ctb * sizeof(*sp));
        const ovhip_alf_ctu *ap = ovhip_shim_alf_params(c, &nc);
        if (ap && nc != (size_t)n_ctb) { fprintf(stderr, "gen_pipe: picture %d: ALF parameters of %zu CTUs\n", k, nc); exit(1); }
        gbuf_push(&out->alf, ap ? (const void *)ap : zero, n_ctb * sizeof(*ap));
        free(zero);
        for (int t = 0; t < 5; ++t) { const int16_t *tp = ovhip_shim_alf_table(c, t, &nt); gbuf_push(&out->tab[t], tp, nt); }
        const ovhip_lmcs_luts *lu = ovhip_shim_lmcs(c);
        ovhip_lmcs_luts none;
        memset(&none, 0, sizeof(none));
        gbuf_push(&out->luts, (d->lmcs && lu) ? lu : &none, sizeof(none));
        ovhip_dbf_offsets offs;
        size_t ne = 0;
        (void)ovhip_rec_dbf_edges(ovhip_shim_recorder(c), 0, &ne, &offs);
        gbuf_push(&out->offs, &offs, sizeof(offs));
        const void *rp[16];
        int32_t map[16];
        const int np = ovhip_shim_ref_pictures(c, rp, 16);
        for (int i = 0; i < 16; ++i) { map[i] = -1; for (int q = 0; i < np && q < k; ++q) if ((const void *)pics[q] == rp[i]) map[i] = q; }
        gbuf_push(&out->refmap, map, 16);
        int32_t pf[4] = { d->lmcs, np, c->lmcs_info.lmcs_enabled_flag, c->lmcs_info.scale_c_flag };
        gbuf_push(&out->pflags, pf, 4);
    stream_only:
        shim_case_end(c, g_pass_shim == 2 ? &out->S2 : &out->S, "pipe picture");
        if (g_pass_shim == 2) {
            /* keys -> picture numbers: the fixture must not hold this run's pointers */
            ovhip_frame_event *ev = (ovhip_frame_event *)g_events.data;
            for (size_t i = 0; i < g_events.n / sizeof(*ev); ++i) {
                if (ev[i].frame < 0 || ev[i].key < 0x10000) continue;
                int64_t idx = -1;
                for (int q = 0; q <= k; ++q) if ((uint64_t)(uintptr_t)pics[q]->frame == ev[i].key) idx = q;
                if (idx < 0) { fprintf(stderr, "gen_pipe: device pass: a frame-level call names a key that is no picture of the stream\n"); exit(1); }
                ev[i].key = (uint64_t)idx;
            }
        }
    }
    if (g_pass_shim == 2) { ovhip_frame_set_trace(NULL, NULL); ovhip_shim_release(c); ovhip_shim_set_dpb(NULL); ovhip_dpb_destroy(dpb); }
}


/* ------------------------------------------------------------------------------------------------ frame threads (device / live passes)
 * What ovdec.c:188-248 + ovthreads.c do for the reference: N sub-decoders, each with its own OVSliceDec and OVCTUDec; the next picture
 * in decoding order goes to the next free one; a picture's readers wait for its CTU rows (dpb.c:1242-1323). */
static void gp_sync_waited(double seconds);
static void
gp_synchro(const OVPicture *const ref_pic, int tl_ctu_x, int tl_ctu_y, int br_ctu_x, int br_ctu_y)
{
    /* ovdpb_synchro_ref_decoded_ctus (dpb.c:1242-1270; static there): wait until the reported-rows mask covers the rectangle */
    const struct PicDecodedCtusInfo *dc = &ref_pic->decoded_ctus;
    double t0 = 0;
    pthread_mutex_lock(dc->ref_mtx);
    for (;;) {
        int ok = 1;
        for (int y = tl_ctu_y; y <= br_ctu_y && ok; ++y)
            for (int x = tl_ctu_x; x <= br_ctu_x && ok; ++x) ok = (int)((dc->mask[y][x >> 6] >> (x & 63)) & 1);
        if (ok) break;
        if (t0 == 0) t0 = gp_now();
        pthread_cond_wait(dc->ref_cnd, dc->ref_mtx);
    }
    pthread_mutex_unlock(dc->ref_mtx);
    if (t0 != 0) gp_sync_waited(gp_now() - t0);
}

struct gp_thread {
    pthread_t th; int id;
    struct gp_seq *s; const struct gp_pic_desc *desc; int n_pic;
    OVSliceDec sl; OVCTUDec *c;
    OVPS ps;                                    /* the active parameter sets of the picture in hand (a kept picture may be decoded by
                                                 * several threads at once: "cont") */
    double t_busy, t_hooks;                     /* seconds with a picture in hand / of them inside the row-end and attach hooks (device waits, flush) */
    double t_sync;                              /* ... waiting for CTU rows of the collocated picture (tmvp_inter_synchronization) */
    double t_shim_hooks, t_shim_device; uint64_t n_shim_calls; uint32_t bands_sent, bands_deferred;      /* the shim's own profile (ovhip_shim_get_profile), "profile" */
    int n_done, err, frames_differing;
    uint64_t samples_differing, mv_cells_differing, mv_cells_compared;
};
static int g_next_pic;
static int *g_readers_left;                     /* pictures still to read picture k (+ 1: its own comparison) */
static __thread struct gp_thread *tls_thread;
static void gp_sync_waited(double seconds) { if (tls_thread) tls_thread->t_sync += seconds; }
static int g_profile, g_noout, g_bands = -1;
/* "timeline": one line per picture of the last repetition on stderr -- which thread took it when, how long its decode call took, how
 * much of that it waited for collocated rows / spent in the row-end hooks, when the in-process comparison ended */
static int g_timeline; static double g_tl_t0;
static struct gp_tl { int thread; double take, decoded, compared, sync, hooks; } g_tl[4096];
static double g_prof_overhead;

static void gp_t_attach(struct OVRCNCtx *const r, const OVFrame *const f, const struct RectEntryInfo *const e, uint8_t l2)
{ const double t0 = gp_now(); __atomic_load_n(&g_attach_inner, __ATOMIC_RELAXED)(r, f, e, l2); if (tls_thread) tls_thread->t_hooks += gp_now() - t0; }
static void gp_t_sao_first(OVCTUDec *const c, const struct RectEntryInfo *const e, uint16_t y)
{ const double t0 = gp_now(); __atomic_load_n(&g_sao_first_inner, __ATOMIC_RELAXED)(c, e, y); if (tls_thread) tls_thread->t_hooks += gp_now() - t0; }
static void gp_t_alf_line(OVCTUDec *const c, const struct RectEntryInfo *const e, uint16_t y)
{ const double t0 = gp_now(); __atomic_load_n(&g_alf_line_inner, __ATOMIC_RELAXED)(c, e, y); if (tls_thread) tls_thread->t_hooks += gp_now() - t0; }

static void gp_reader_done(int k);
static int gp_out_threads_on(void);
static void gp_out_push_fwd(int k);

static void
gp_compare(struct gp_thread *t, const struct gp_seq *s, int k)
{
    const OVPicture *a = g_kept[k].pic, *b = g_kept[k].ref_pic;
    uint64_t nd = 0, nm = 0, nc = 0;
    if (g_pass_shim == 3 && !g_noout) {
        for (int p = 0; p < 3; ++p) {
            const size_t n = (size_t)(p ? s->w / 2 : s->w) * (size_t)(p ? s->h / 2 : s->h);
            const uint16_t *x = (const uint16_t *)a->frame->data[p], *y = (const uint16_t *)b->frame->data[p];
            if (memcmp(x, y, n * 2)) for (size_t i = 0; i < n; ++i) nd += x[i] != y[i];
        }
    }
    /* the collocated motion field as its readers see it (tmvp_store_mv, drv_lines.c:270-330; load_ctb_tmvp, drv_mvp.c:298-345): per CTU 32
     * direction words (column of 4-sample units -> bit 1 + row) and 16 x 16 cells of 8x8 samples, a cell being meaningful when the
     * direction bit of its top-left unit is set.  Cells whose bit is clear hold whatever the OVCTUDec's CTU-local array held before --
     * another picture's vectors, which depend on the pictures that OVCTUDec happened to decode -- and nobody reads them. */
    const size_t n_ctb = (size_t)s->nb_ctb_w * s->nb_ctb_h;
    const int pln_stride = 16 * s->nb_ctb_w;
    for (int l = 0; l < 2; ++l) {
        const struct MVPlane *pa = l ? &a->mv_plane1 : &a->mv_plane0, *pb = l ? &b->mv_plane1 : &b->mv_plane0;
        nm += memcmp(pa->dirs, pb->dirs, n_ctb * 32 * 8) != 0;
        for (int cy = 0; cy < s->nb_ctb_h; ++cy) for (int cx = 0; cx < s->nb_ctb_w; ++cx) {
            const uint64_t *dirs = pb->dirs + (size_t)(cx + cy * s->nb_ctb_w) * 32;
            for (int uy = 0; uy < 16; ++uy) for (int ux = 0; ux < 16; ++ux) {
                if (!((dirs[2 * ux] >> (2 * uy + 1)) & 1)) continue;          /* vfield: bit 0 is the row above the CTU */
                const size_t i = (size_t)(cx + cy * pln_stride) * 16 + (size_t)uy * pln_stride + ux;
                const int df = pa->mvs[i].x != pb->mvs[i].x || pa->mvs[i].y != pb->mvs[i].y || pa->mvs[i].ref_idx != pb->mvs[i].ref_idx;
                if (df && nm < 8 && getenv("GP_TRACE_MV")) fprintf(stderr, "    pic %d list %d cell %zu: (%d, %d, ref %d) vs reference (%d, %d, ref %d)\n", k, l, i, pa->mvs[i].x, pa->mvs[i].y, pa->mvs[i].ref_idx, pb->mvs[i].x, pb->mvs[i].y, pb->mvs[i].ref_idx);
                nm += df; nc++;
            }
        }
    }
    t->samples_differing += nd; t->mv_cells_differing += nm; t->mv_cells_compared += nc; t->frames_differing += nd != 0;
    if (nd || nm) fprintf(stderr, "gen_pipe: picture %d (POC %d): %llu samples, %llu collocated-motion entries differ from the reference pass\n", k, a->poc,
                          (unsigned long long)nd, (unsigned long long)nm);
}

static void
gp_decode_kept(struct gp_thread *t, int k)
{
    struct gp_kept *kp = &g_kept[k];
    const struct gp_pic_desc *d = &t->desc[k];
    memset(&t->ps, 0, sizeof(t->ps));
    if (decinit_update_params(&t->ps, &kp->nvcl) < 0) { fprintf(stderr, "gen_pipe: decinit_update_params failed\n"); exit(1); }
    const int n_entries = g_tile_cols * g_tile_rows;
    const size_t per_entry = (GP_PAYLOAD / n_entries) & ~(size_t)63;
    for (int i = 0; i <= n_entries; ++i) t->ps.sh_info.rbsp_entry[i] = kp->payload + i * per_entry;
    t->sl.pic = kp->pic; t->sl.active_params = &t->ps; t->sl.slice_type = d->slice_type;
    slicedec_init_lines(&t->sl, &t->ps);
    g_dmvr_pos = kp->dm0;
    for (int i = 0; i < n_entries; ++i) {
        slicedec_update_entry_decoder(&t->sl, t->c);
        slicedec_decode_rect_entry(&t->sl, t->c, &t->ps, i);
    }
    ovdpb_report_decoded_frame(kp->pic);
    if (g_pass_shim == 2 && g_dmvr_pos != kp->dm1) { fprintf(stderr, "gen_pipe: picture %d made %zu DMVR calls, the reference pass %zu\n", k, g_dmvr_pos - kp->dm0, kp->dm1 - kp->dm0); exit(1); }
    if (g_pass_shim == 2) ovhip_shim_flush_pending(t->c);
    const int e = ovhip_shim_last_error(t->c);
    if (e) { fprintf(stderr, "gen_pipe: picture %d: the shim latched %d\n", k, e); if (!t->err) t->err = e; }
    if (g_timeline) { g_tl[k].decoded = gp_now() - g_tl_t0; }
    if (!gp_out_threads_on()) {
        gp_compare(t, t->s, k);
        if (g_timeline) { g_tl[k].compared = gp_now() - g_tl_t0; }
    }
    for (int i = 0; i < d->n0; ++i) gp_reader_done(d->l0[i]);
    for (int i = 0; i < d->n1; ++i) gp_reader_done(d->l1[i]);
    if (gp_out_threads_on()) gp_out_push_fwd(k); else gp_reader_done(k);
    t->n_done++;
}

/* ---- the decoder's frame pool (ovframepool.c): a picture gets its OVFrame when a frame thread takes it (ovdpb_init_picture, when the
 * slice header has been read) and gives it back when the last picture that references it and the output are done with it -- the NEXT
 * picture to be taken gets that very OVFrame.  The shim keys the device DPB by the OVFrame pointer + (cvs, POC): this is what makes
 * a recycled key meet its previous owner's slot (include/ovvc_hip.h, ovhip_dpb_begin_tag). */
static pthread_mutex_t g_take_mtx = PTHREAD_MUTEX_INITIALIZER;
static OVFrame *g_frame_pool[GP_MAX_PIC]; static int g_n_frame_pool, g_frames_made, g_frames_recycled;
static int g_no_release;                         /* "norelease": the decoder never tells the shim that a frame was dropped (ovhip_shim_frame_released is optional) */
static int g_cont = 1;                           /* "cont C": the stream C times BACK TO BACK as one sequence of C x pics pictures (C coded video sequences: the frame threads
                                                  * take the next copy's pictures while the tail of the one before still decodes -- the steady state a short stream's start-up and
                                                  * tail hide; one reference pass serves every copy) */
static int g_reps = 1, g_rep;                    /* "reps R": the stream R times on the same frame threads (contexts, jobs, recorders warm); the last one is timed */
static pthread_barrier_t g_bar;

static OVFrame *
gp_frame_get(const struct gp_seq *s)
{
    if (g_n_frame_pool) { g_frames_recycled++; return g_frame_pool[--g_n_frame_pool]; }
    OVFrame *f = calloc(1, sizeof(*f));
    f->width = s->w; f->height = s->h;
    f->linesize[0] = (size_t)s->w * 2; f->linesize[1] = f->linesize[2] = (size_t)(s->w / 2) * 2;
    for (int k = 0; k < 3; ++k) {                 /* 16 rows of slack on both sides, as gp_new_picture */
        const size_t wk = (size_t)(k ? s->w / 2 : s->w), hk = (size_t)(k ? s->h / 2 : s->h);
        f->data[k] = (uint8_t *)calloc(wk * (hk + 32), 2) + wk * 16 * 2;
    }
    g_frames_made++;
    return f;
}

static void
gp_reader_done(int k)
{
    if (__atomic_sub_fetch(&g_readers_left[k], 1, __ATOMIC_ACQ_REL)) return;
    /* the decoder dropped its last reference to the frame (ovframe_unref reaching zero): the one-line call INTEGRATION.md section 3 adds,
     * then the frame is the pool's again */
    OVFrame *f = g_kept[k].pic->frame;
    if (g_pass_shim == 3 && !g_no_release) ovhip_shim_frame_released(f);
    pthread_mutex_lock(&g_take_mtx);
    g_frame_pool[g_n_frame_pool++] = f;
    pthread_mutex_unlock(&g_take_mtx);
}

/* "outthreads N": the comparison of a decoded picture with the reference pass (and the release of its frame) is done by N OUTPUT
 * threads, not by the frame thread that decoded it -- as in the reference's own application, where the frame threads never touch a
 * picture's output (dectest's main thread takes the frames from ovdec_receive_picture and writes them, dectest.c:372-409).  The timed
 * region ends when every picture has been compared.  Default 0: the frame thread compares (what every figure before round 6 measured). */
static int g_out_threads;
static struct { pthread_mutex_t mtx; pthread_cond_t cnd; int q[4096]; int head, tail, done, quit; } g_outq = { PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER };
static void
gp_out_push(int k)
{
    pthread_mutex_lock(&g_outq.mtx);
    g_outq.q[g_outq.tail++ & 4095] = k;
    pthread_cond_broadcast(&g_outq.cnd);
    pthread_mutex_unlock(&g_outq.mtx);
}
static void *
gp_out_worker(void *arg)
{
    struct gp_thread *t = (struct gp_thread *)arg;          /* (its counters only) */
    for (;;) {
        pthread_mutex_lock(&g_outq.mtx);
        while (g_outq.head == g_outq.tail && !g_outq.quit) pthread_cond_wait(&g_outq.cnd, &g_outq.mtx);
        if (g_outq.head == g_outq.tail) { pthread_mutex_unlock(&g_outq.mtx); return NULL; }
        const int k = g_outq.q[g_outq.head++ & 4095];
        pthread_mutex_unlock(&g_outq.mtx);
        gp_compare(t, t->s, k);
        if (g_timeline) g_tl[k].compared = gp_now() - g_tl_t0;
        gp_reader_done(k);
        pthread_mutex_lock(&g_outq.mtx);
        g_outq.done++;
        pthread_cond_broadcast(&g_outq.cnd);
        pthread_mutex_unlock(&g_outq.mtx);
    }
}

static void *
gp_worker(void *arg)
{
    struct gp_thread *t = (struct gp_thread *)arg;
    tls_thread = t;
    for (int rep = 0; rep < g_reps; ++rep) {
        pthread_barrier_wait(&g_bar);                                   /* the main thread has made the repetition's pictures */
        t->t_busy = t->t_hooks = t->t_sync = 0; t->n_done = 0;
        if (g_profile && g_pass_shim == 3) { ovhip_shim_profile pr; (void)ovhip_shim_get_profile(t->c, &pr, 1); }
        for (;;) {
            pthread_mutex_lock(&g_take_mtx);
            const int k = g_next_pic < t->n_pic ? g_next_pic++ : -1;
            if (k >= 0) g_kept[k].pic->frame = gp_frame_get(t->s);
            pthread_mutex_unlock(&g_take_mtx);
            if (k < 0) break;
            const double t0 = gp_now(), sy0 = t->t_sync, hk0 = t->t_hooks;
            if (g_timeline) { g_tl[k].thread = t->id; g_tl[k].take = t0 - g_tl_t0; }
            gp_decode_kept(t, k);
            t->t_busy += gp_now() - t0;
            if (g_timeline) { g_tl[k].sync = t->t_sync - sy0; g_tl[k].hooks = t->t_hooks - hk0; }
        }
        ovhip_shim_band_stats(t->c, &t->bands_sent, &t->bands_deferred);
        if (g_profile && g_pass_shim == 3) {
            ovhip_shim_profile pr;
            if (ovhip_shim_get_profile(t->c, &pr, 0) == 0) { t->t_shim_hooks = pr.seconds_in_hooks; t->t_shim_device = pr.seconds_device; t->n_shim_calls = pr.n_calls; g_prof_overhead = pr.seconds_overhead_per_call; }
        }
        pthread_barrier_wait(&g_bar);
    }
    return NULL;
}

static int gp_out_threads_on(void) { return g_out_threads > 0 && g_pass_shim == 3; }
static void gp_out_push_fwd(int k) { gp_out_push(k); }

/* returns the wall time of the (last repetition of the) pass; the per-thread sums land in *tot */
static double
run_stream_threads(struct gp_seq *s, const struct gp_pic_desc *desc, int n_pic, int n_threads, struct gp_thread *tot)
{
    ovhip_dpb *dpb = NULL;
    if (g_pass_shim == 2) {
        ovhip_dpb_ops ops;
        memset(&ops, 0, sizeof(ops));
        ops.pic_alloc = fm_alloc; ops.pic_free = fm_free; ops.copy_start = fm_copy_start; ops.copy_wait = fm_copy_wait;
        if (ovhip_dpb_create_ex(&dpb, 1, &ops)) { fprintf(stderr, "gen_pipe: ovhip_dpb_create_ex failed\n"); exit(1); }
        ovhip_shim_set_dpb(dpb);
    }
    struct gp_thread *th = calloc(n_threads, sizeof(*th));
    for (int i = 0; i < n_threads; ++i) {
        th[i].id = i; th[i].s = s; th[i].desc = desc; th[i].n_pic = n_pic;
        gp_alloc_lines(&th[i].sl, s);
        th[i].c = gp_new_ctudec(s);
    }
    pthread_barrier_init(&g_bar, NULL, n_threads + 1);
    g_readers_left = calloc(n_pic, sizeof(int));
    g_frames_made = g_frames_recycled = 0;
    for (int i = 0; i < n_threads; ++i) if (pthread_create(&th[i].th, NULL, gp_worker, &th[i])) { perror("pthread_create"); exit(1); }
    struct gp_thread *oth = NULL;
    if (gp_out_threads_on()) {
        oth = calloc(g_out_threads, sizeof(*oth));
        g_outq.head = g_outq.tail = g_outq.done = g_outq.quit = 0;
        for (int i = 0; i < g_out_threads; ++i) { oth[i].id = 1000 + i; oth[i].s = s; if (pthread_create(&oth[i].th, NULL, gp_out_worker, &oth[i])) { perror("pthread_create"); exit(1); } }
    }
    double wall = 0;
    for (g_rep = 0; g_rep < g_reps; ++g_rep) {
        /* the pictures exist as objects before any thread runs (the decoder's DPB makes them when it reads the slice header:
         * ovdpb_init_picture); their frames come from the pool when a thread takes them.  A repetition is a new coded video sequence:
         * the same POCs are other pictures (pic_tag in the shim) */
        for (int k = 0; k < n_pic; ++k) {
            const struct gp_pic_desc *d = &desc[k];
            OVPicture *p = g_kept[k].pic = gp_new_picture(s, d->poc);
            for (int q = 0; q < 3; ++q) { const size_t wk = (size_t)(q ? s->w / 2 : s->w); free(p->frame->data[q] - wk * 16 * 2); }
            free(p->frame); p->frame = NULL;
            p->cvs_id = (uint16_t)(g_rep * g_cont + k / (n_pic / g_cont));
            atomic_init(&p->idx_function, 1);
            p->ovdpb_frame_synchro[1] = gp_synchro;
            OVPicture *l0[2] = { d->n0 > 0 ? g_kept[d->l0[0]].pic : NULL, d->n0 > 1 ? g_kept[d->l0[1]].pic : NULL };
            OVPicture *l1[2] = { d->n1 > 0 ? g_kept[d->l1[0]].pic : NULL, d->n1 > 1 ? g_kept[d->l1[1]].pic : NULL };
            gp_set_refs(p, l0, d->n0, l1, d->n1, d->tmvp, d->col_from_l0);
            g_readers_left[k] = 1;
        }
        for (int k = 0; k < n_pic; ++k) {
            const struct gp_pic_desc *d = &desc[k];
            for (int i = 0; i < d->n0; ++i) g_readers_left[d->l0[i]]++;
            for (int i = 0; i < d->n1; ++i) g_readers_left[d->l1[i]]++;
        }
        g_next_pic = 0;
        const double t0 = gp_now();
        g_tl_t0 = t0;
        if (g_timeline) fprintf(stderr, "repetition %d starts at %.3f ms (CLOCK_MONOTONIC)\n", g_rep, 1e3 * t0);
        pthread_barrier_wait(&g_bar);
        pthread_barrier_wait(&g_bar);
        if (gp_out_threads_on()) {                                   /* the timed region ends when every picture has been compared */
            pthread_mutex_lock(&g_outq.mtx);
            while (g_outq.done < (g_rep + 1) * n_pic) pthread_cond_wait(&g_outq.cnd, &g_outq.mtx);
            pthread_mutex_unlock(&g_outq.mtx);
        }
        wall = gp_now() - t0;
        if (g_timeline && g_rep == g_reps - 1)
            for (int k = 0; k < n_pic; ++k)
                fprintf(stderr, "timeline pic %3d poc %3d type %d thr %2d take %7.2f decoded %7.2f compared %7.2f ms | in decode: waited for collocated rows %6.2f, row-end hooks %6.2f\n",
                        k, desc[k].poc, desc[k].slice_type, g_tl[k].thread, 1e3 * g_tl[k].take, 1e3 * g_tl[k].decoded, 1e3 * g_tl[k].compared, 1e3 * g_tl[k].sync, 1e3 * g_tl[k].hooks);
        for (int k = 0; k < n_pic; ++k) {
            OVPicture *p = g_kept[k].pic;
            for (int i = 0; i < s->nb_ctb_h; ++i) free(p->decoded_ctus.mask[i]);
            free(p->decoded_ctus.mask);
            free(p->mv_plane0.mvs); free(p->mv_plane0.dirs); free(p->mv_plane1.mvs); free(p->mv_plane1.dirs);
            free(p);
            g_kept[k].pic = NULL;
        }
    }
    for (int i = 0; i < n_threads; ++i) pthread_join(th[i].th, NULL);
    pthread_barrier_destroy(&g_bar);
    memset(tot, 0, sizeof(*tot));
    if (oth) {
        pthread_mutex_lock(&g_outq.mtx); g_outq.quit = 1; pthread_cond_broadcast(&g_outq.cnd); pthread_mutex_unlock(&g_outq.mtx);
        for (int i = 0; i < g_out_threads; ++i) {
            pthread_join(oth[i].th, NULL);
            tot->frames_differing += oth[i].frames_differing; tot->samples_differing += oth[i].samples_differing;
            tot->mv_cells_differing += oth[i].mv_cells_differing; tot->mv_cells_compared += oth[i].mv_cells_compared;
        }
        free(oth);
    }
    for (int i = 0; i < n_threads; ++i) {
        tot->t_sync += th[i].t_sync; tot->t_shim_hooks += th[i].t_shim_hooks; tot->t_shim_device += th[i].t_shim_device; tot->n_shim_calls += th[i].n_shim_calls;
        tot->bands_sent += th[i].bands_sent; tot->bands_deferred += th[i].bands_deferred;
        tot->t_busy += th[i].t_busy; tot->t_hooks += th[i].t_hooks; tot->n_done += th[i].n_done; tot->frames_differing += th[i].frames_differing;
        tot->samples_differing += th[i].samples_differing; tot->mv_cells_differing += th[i].mv_cells_differing; tot->mv_cells_compared += th[i].mv_cells_compared;
        if (th[i].err && !tot->err) tot->err = th[i].err;
        ovhip_shim_release(th[i].c);                  /* the frame thread's device context and job; the last one takes the shim's DPB with it */
    }
    if (g_pass_shim == 2) { ovhip_shim_set_dpb(NULL); ovhip_dpb_destroy(dpb); }
    free(th);
    while (g_n_frame_pool) {
        OVFrame *f = g_frame_pool[--g_n_frame_pool];
        for (int q = 0; q < 3; ++q) { const size_t wk = (size_t)(q ? s->w / 2 : s->w); free(f->data[q] - wk * 16 * 2); }
        free(f);
    }
    free(g_readers_left); g_readers_left = NULL;
    return wall;
}

/* a crash inside the reference on a stream it was never written for (the parse is a random walk): say where, exit 3 */
static void
gp_on_segv(int sig)
{
    void *bt[48];
    const int n = backtrace(bt, 48);
    static const char msg[] = "gen_pipe: signal inside the reference decoder; backtrace:\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(bt, n, 2);
    _exit(3);
}

int
gp_main(int argc, char **argv)
{
    signal(SIGSEGV, gp_on_segv); signal(SIGBUS, gp_on_segv); signal(SIGFPE, gp_on_segv); signal(SIGABRT, gp_on_segv);
    const char *dir = argc > 1 ? argv[1] : "../tests/golden";
    int want_shim = 0, want_dev = 0, want_live = 0, want_time = 0, variant = 0, W = 416, H = 240, dqp = 0, n_pic = 5, gop_size = 0;
    uint32_t seed = 0x266 + 31337;
    const char *name = "pipe";
    /* gen_pipe <dir> [shim | device | live | simd] [threads <n>[,<m>...]] [reps <n>] [cont <copies back to back>] [gop <8|16|32>] [profile] [noout] [norelease] [name <fixture name>] [seed <n>] [variant <0|1>] [qp <delta on every picture's QP>]
     *          [size <w> <h>] [pics <1..257>] [time] */
    for (int i = 2; i < argc; ++i) {
        if (!strcmp(argv[i], "shim")) want_shim = 1;
        else if (!strcmp(argv[i], "device")) want_shim = want_dev = 1;
        else if (!strcmp(argv[i], "reps") && i + 1 < argc) g_reps = atoi(argv[++i]);
        else if (!strcmp(argv[i], "cont") && i + 1 < argc) g_cont = atoi(argv[++i]);
        else if (!strcmp(argv[i], "gop") && i + 1 < argc) gop_size = atoi(argv[++i]);
        else if (!strcmp(argv[i], "noisp")) g_no_isp = 1;
        else if (!strcmp(argv[i], "norelease")) g_no_release = 1;
        else if (!strcmp(argv[i], "profile")) g_profile = 1;      /* live: the shim's own split of a frame thread's time (ovhip_shim_set_profile) */
        else if (!strcmp(argv[i], "noout")) g_noout = 1;          /* live: OVHIP_OUT_NONE -- the pictures stay on the device (no copy into the OVFrame, frames not compared) */
        else if (!strcmp(argv[i], "timeline")) g_timeline = 1;
        else if (!strcmp(argv[i], "allow64x2")) g_allow_64x2 = 1;
        else if (!strcmp(argv[i], "outthreads") && i + 1 < argc) g_out_threads = atoi(argv[++i]);
        else if (!strcmp(argv[i], "bands") && i + 1 < argc) g_bands = atoi(argv[++i]);      /* CTU rows per band (ovhip_shim_set_bands); default: the shim's */
        else if (!strcmp(argv[i], "live")) want_live = 1;      /* the shim on the real device, on frame threads; compares in process, prints a JSON line */
        else if (!strcmp(argv[i], "threads") && i + 1 < argc) {
            for (const char *q = argv[++i]; *q && g_n_thread_list < 16;) { g_thread_list[g_n_thread_list++] = atoi(q); while (*q && *q != ',') ++q; if (*q) ++q; }
            g_threads = g_n_thread_list ? g_thread_list[0] : 0;
        }
        else if (!strcmp(argv[i], "simd")) g_simd = 1;        /* the reference pass through the reference's SSE4.1 / AVX2 back-end (ref_common.h) */
        else if (!strcmp(argv[i], "name") && i + 1 < argc) name = argv[++i];
        else if (!strcmp(argv[i], "seed") && i + 1 < argc) seed = (uint32_t)strtoul(argv[++i], NULL, 0);
        else if (!strcmp(argv[i], "variant") && i + 1 < argc) variant = atoi(argv[++i]);
        else if (!strcmp(argv[i], "qp") && i + 1 < argc) dqp = atoi(argv[++i]);
        else if (!strcmp(argv[i], "pics") && i + 1 < argc) n_pic = atoi(argv[++i]);
        else if (!strcmp(argv[i], "null")) g_null_shim = 1;      /* with "time shim": the installed slots without a recorder = the parse alone */
        else if (!strcmp(argv[i], "time")) want_time = g_time_only = 1;       /* reference pass only; prints pictures and seconds inside the slice decoder */
        else if (!strcmp(argv[i], "size") && i + 2 < argc) { W = atoi(argv[i + 1]); H = atoi(argv[i + 2]); i += 2; }
        else if (!strcmp(argv[i], "tiles") && i + 2 < argc) { g_tile_cols = atoi(argv[i + 1]); g_tile_rows = atoi(argv[i + 2]); i += 2; }
        else { fprintf(stderr, "gen_pipe: unknown argument %s\n", argv[i]); return 2; }
    }
    if (W % 8 || H % 8 || W < 136 || H < 136 || W > 4096 || H > 2304 || n_pic < 1 || n_pic > GP_MAX_PIC) { fprintf(stderr, "gen_pipe: size / pics\n"); return 2; }
    if (want_live && !g_threads) { g_threads = 1; g_thread_list[0] = 1; g_n_thread_list = 1; }
    if (gop_size && ((gop_size != 8 && gop_size != 16 && gop_size != 32) || (!g_threads && !want_time))) { fprintf(stderr, "gen_pipe: gop 8 | 16 | 32, with threads or time\n"); return 2; }
#ifdef OVVC_HIP_CALLER_PATCH
    /* the patched caller never comes through rcn_dmvr_mv_refine: nothing can be fed back from the reference pass, so the modes that
     * record without a device (whose later pictures' parse needs the refined vectors) do not exist here */
    if (!want_live && !want_time) { fprintf(stderr, "gen_pipe (caller patch): live and time modes only\n"); return 2; }
#endif
    for (int i = 0; i < g_n_thread_list; ++i) if (g_thread_list[i] < 1 || g_thread_list[i] > 64) { fprintf(stderr, "gen_pipe: threads\n"); return 2; }
    if (g_threads < 0 || g_threads > 64 || (g_threads && !want_live && !want_dev) || (g_threads && want_time)) { fprintf(stderr, "gen_pipe: threads\n"); return 2; }
    if (g_threads) g_kept = calloc(n_pic, sizeof(*g_kept));
    while (g_payload_bytes < (size_t)W * H * 2) g_payload_bytes <<= 1;          /* 16 bits per sample: far above any slice's need */
    if (!g_threads && posix_memalign((void **)&g_payload, 64, GP_PAYLOAD)) abort();
    /* decoding order of a hierarchical GOP of 8: I0, B8 (two lists to the I picture), B4 (between them: DMVR / BDOF / SMVD have a past
     * and a future reference, TMVP from B8), B2, P6; with "pics 9" the rest of the GOP: b1, b3, b5, b7 */
    static struct gp_pic_desc gop[GP_MAX_PIC + 1] = {
        { .poc = 0, .slice_type = 2, .qp = 30, .lmcs = 1 },
        { .poc = 8, .slice_type = 0, .qp = 33, .l0 = { 0 }, .n0 = 1, .l1 = { 0 }, .n1 = 1, .tmvp = 0, .col_from_l0 = 1, .lmcs = 1 },
        { .poc = 4, .slice_type = 0, .qp = 35, .l0 = { 0, 1 }, .n0 = 2, .l1 = { 1, 0 }, .n1 = 2, .tmvp = 1, .col_from_l0 = 0, .lmcs = 1 },
        { .poc = 2, .slice_type = 0, .qp = 37, .l0 = { 0, 2 }, .n0 = 2, .l1 = { 2, 1 }, .n1 = 2, .tmvp = 1, .col_from_l0 = 0, .lmcs = 0 },
        { .poc = 6, .slice_type = 1, .qp = 37, .l0 = { 2, 0 }, .n0 = 2, .n1 = 0, .tmvp = 1, .col_from_l0 = 1, .lmcs = 1 },
        { .poc = 1, .slice_type = 0, .qp = 38, .l0 = { 0, 3 }, .n0 = 2, .l1 = { 3, 2 }, .n1 = 2, .tmvp = 1, .col_from_l0 = 0, .lmcs = 1 },
        { .poc = 3, .slice_type = 0, .qp = 38, .l0 = { 3, 0 }, .n0 = 2, .l1 = { 2, 4 }, .n1 = 2, .tmvp = 1, .col_from_l0 = 1, .lmcs = 1 },
        { .poc = 5, .slice_type = 0, .qp = 38, .l0 = { 2, 3 }, .n0 = 2, .l1 = { 4, 1 }, .n1 = 2, .tmvp = 1, .col_from_l0 = 0, .lmcs = 1 },
        { .poc = 7, .slice_type = 0, .qp = 38, .l0 = { 4, 2 }, .n0 = 2, .l1 = { 1, 4 }, .n1 = 2, .tmvp = 1, .col_from_l0 = 1, .lmcs = 1 },
    };
    /* "pics" beyond 9: further GOPs of 8 with the same structure, each from the key picture before it: picture j (1..8) of GOP g is
     * picture j of the table with its references moved along -- entry 0 (the I picture) becomes the previous GOP's key picture, entry
     * e > 0 becomes e + 8 g -- and 8 g added to its POC; key pictures after the first have a motion field to take TMVP from */
    for (int k = 9; k < n_pic; ++k) {
        const int g = (k - 1) / 8, j = (k - 1) % 8 + 1;
        gop[k] = gop[j];
        gop[k].poc += 8 * g;
        for (int i = 0; i < 2; ++i) {
            gop[k].l0[i] = gop[j].l0[i] ? gop[j].l0[i] + 8 * g : 1 + 8 * (g - 1);
            gop[k].l1[i] = gop[j].l1[i] ? gop[j].l1[i] + 8 * g : 1 + 8 * (g - 1);
        }
        if (j == 1) gop[k].tmvp = 1;
    }
    if (gop_size) {
        /* "gop G" (live / time modes): a hierarchical-B random-access stream of GOPs of G = 8 / 16 / 32 pictures in depth-first decoding
         * order, as the JVET CTC random-access configuration codes them (GOP 32: 32 16 8 4 2 1 3 6 5 7 12 ...): the key picture from the
         * previous key picture, every other picture from the two pictures it lies between -- up to G / 2 pictures of the last layer
         * are independent of each other, which is the parallelism frame threads live on.  Pictures beyond the last whole GOP are dropped. */
        int n = 1, prev_key = 0;
        struct { int lo, hi, lo_poc, hi_poc, depth; } stack[64];
        while (n + gop_size <= n_pic) {
            const int base = gop[prev_key].poc, key = n;
            gop[n++] = (struct gp_pic_desc){ .poc = base + gop_size, .slice_type = 0, .qp = 33, .l0 = { prev_key }, .n0 = 1, .l1 = { prev_key }, .n1 = 1,
                                             .tmvp = prev_key != 0, .col_from_l0 = 1, .lmcs = 1 };
            int sp = 0;
            stack[sp].lo = prev_key; stack[sp].hi = key; stack[sp].lo_poc = base; stack[sp].hi_poc = base + gop_size; stack[sp].depth = 1; ++sp;
            while (sp) {
                --sp;
                const int lo = stack[sp].lo, hi = stack[sp].hi, lp = stack[sp].lo_poc, hp = stack[sp].hi_poc, d = stack[sp].depth;
                if (hp - lp < 2) continue;
                const int mid = n;
                gop[n++] = (struct gp_pic_desc){ .poc = (lp + hp) / 2, .slice_type = 0, .qp = 34 + d, .l0 = { lo, hi }, .n0 = 2, .l1 = { hi, lo }, .n1 = 2,
                                                 .tmvp = 1, .col_from_l0 = d & 1, .lmcs = 1 };
                /* depth first, the earlier half first: push the later half below it */
                stack[sp].lo = mid; stack[sp].hi = hi; stack[sp].lo_poc = (lp + hp) / 2; stack[sp].hi_poc = hp; stack[sp].depth = d + 1; ++sp;
                stack[sp].lo = lo; stack[sp].hi = mid; stack[sp].lo_poc = lp; stack[sp].hi_poc = (lp + hp) / 2; stack[sp].depth = d + 1; ++sp;
            }
            prev_key = key;
        }
        n_pic = n;
    }
    for (int k = 0; k < n_pic; ++k) gop[k].qp += dqp;
    struct gp_out out;
    memset(&out, 0, sizeof(out));
    out.frames.type = T_U16; out.info.type = T_I32; out.sao.type = T_U8; out.alf.type = T_U8; out.luts.type = T_U8; out.offs.type = T_U8;
    out.refmap.type = T_I32; out.pflags.type = T_I32;
    for (int t = 0; t < 5; ++t) out.tab[t].type = T_I16;
    shim_stream_init(&out.S); shim_stream_init(&out.S2);

    static struct gp_seq seq;
    if (g_threads) {
        /* reference pass, then ONE pass on frame threads -- dry frames ("device threads N": the shim, the DPB's state machine and this
         * harness under several threads, no GPU) or live -- over the slice data and headers the reference pass kept */
        g_seed = 0x266 + 4242;
        seq_init(&seq, W, H, variant);
        fprintf(stderr, "gen_pipe: reference pass\n");
        const double tr0 = gp_now();
        run_stream(&seq, gop, n_pic, seed, &out);
        const double t_ref = gp_now() - tr0;
        if (g_isp_64x2 && !(g_allow_64x2 && want_live)) { fprintf(stderr, "gen_pipe: the stream holds %d 64x2 ISP partitions (reference result undefined): pick another seed\n", g_isp_64x2); return 1; }
        const int n_64x2_ref = g_isp_64x2;
        g_pass_shim = want_live ? 3 : 2;
        if (want_live && g_profile) ovhip_shim_set_profile(1);
        if (want_live && g_noout) ovhip_shim_set_output(OVHIP_OUT_NONE);
        if (g_bands >= 0) ovhip_shim_set_bands(g_bands);
        int bad = 0;
        const int n_one = n_pic;
        struct gp_pic_desc *gop_all = gop;
        if (g_cont > 1) {
            /* the kept pictures and their descriptions, C times: copy c's pictures reference copy c's pictures */
            g_kept = realloc(g_kept, (size_t)n_one * g_cont * sizeof(*g_kept));
            gop_all = calloc((size_t)n_one * g_cont, sizeof(*gop_all));
            if (!g_kept || !gop_all) { fprintf(stderr, "gen_pipe: cont: out of memory\n"); return 1; }
            for (int c = 0; c < g_cont; ++c)
                for (int k = 0; k < n_one; ++k) {
                    struct gp_kept *kp = &g_kept[c * n_one + k];
                    if (c) *kp = g_kept[k];
                    kp->nvcl.ph = &kp->ph; kp->nvcl.sh = &kp->sh;          /* (the array has moved) */
                    struct gp_pic_desc d = gop[k];
                    for (int q = 0; q < d.n0; ++q) d.l0[q] += c * n_one;
                    for (int q = 0; q < d.n1; ++q) d.l1[q] += c * n_one;
                    gop_all[c * n_one + k] = d;
                }
            n_pic = n_one * g_cont;
        }
        for (int li = 0; li < g_n_thread_list; ++li) {
            g_threads = g_thread_list[li];
            fprintf(stderr, "gen_pipe: %s pass, %d frame thread%s\n", want_live ? "live" : "device (dry)", g_threads, g_threads > 1 ? "s" : "");
            struct gp_thread tot;
            const double wall = run_stream_threads(&seq, gop_all, n_pic, g_threads, &tot);
            printf("{\"mode\": \"%s\", \"frame_threads\": %d, \"pictures\": %d, \"width\": %d, \"height\": %d, \"seconds\": %.6f, \"pictures_per_second\": %.3f, "
                   "\"pictures_decoded\": %d, \"shim_error\": %d, \"frames_differing\": %d, \"samples_differing\": %llu, \"collocated_motion_entries_differing\": %llu, \"collocated_motion_entries_compared\": %llu, "
                   "\"dmvr_calls\": %zu, \"thread_seconds_with_a_picture\": %.6f, \"thread_seconds_in_row_end_and_attach_hooks\": %.6f, "
                   "\"reference_pass_seconds_inside_slicedec\": %.6f, \"reference_pass_seconds\": %.6f, \"repetitions\": %d, \"host_frames_made\": %d, \"host_frames_recycled\": %d, \"output\": \"%s\", "
                   "\"thread_seconds_waiting_for_collocated_rows\": %.6f, \"shim_profile\": %d, \"thread_seconds_in_shim_hooks\": %.6f, \"thread_seconds_in_shim_device_half\": %.6f, \"shim_hook_calls\": %llu, \"shim_profile_overhead_seconds_per_call\": %.3e, \"copies_back_to_back\": %d, \"bands_sent\": %u, \"bands_left_to_a_later_hook\": %u, \"coding_units_split_into_64x2_isp_partitions\": %d, \"output_threads\": %d}\n",
                   want_live ? "live" : "device_dry_threads", g_threads, n_pic, W, H, wall, n_pic / wall, tot.n_done, tot.err, tot.frames_differing,
                   (unsigned long long)tot.samples_differing, (unsigned long long)tot.mv_cells_differing, (unsigned long long)tot.mv_cells_compared, g_dmvr_log.n / 12, tot.t_busy, tot.t_hooks,
                   g_decode_seconds_pass[0], t_ref, g_reps, g_frames_made, g_frames_recycled, g_noout ? "none" : "planes into the OVFrame",
                   tot.t_sync, g_profile, tot.t_shim_hooks, tot.t_shim_device, (unsigned long long)tot.n_shim_calls, g_prof_overhead, g_cont, tot.bands_sent, tot.bands_deferred, n_64x2_ref, gp_out_threads_on() ? g_out_threads : 0);
            fflush(stdout);
            /* (with 64x2 partitions in the stream the reference pass is not a reference: only errors count) */
            bad |= tot.err || tot.n_done != n_pic || (!n_64x2_ref && (tot.samples_differing || tot.mv_cells_differing));
        }
        return bad;
    }
    ovhip_shim_set_bands(g_bands >= 0 ? g_bands : 0);     /* (the committed device-mode fixtures hold the whole-picture call sequence) */
    for (g_pass_shim = 0; g_pass_shim <= want_shim + want_dev; ++g_pass_shim) {
        g_seed = 0x266 + 4242;
        g_dmvr_pos = 0;
        seq_init(&seq, W, H, variant);
        fprintf(stderr, "gen_pipe: %s pass\n", g_pass_shim == 2 ? "device (dry)" : g_pass_shim ? "shim" : "reference");
        run_stream(&seq, gop, n_pic, seed, &out);
    }
    if (want_time) {
        printf("{\"pictures\": %d, \"width\": %d, \"height\": %d, \"seconds\": %.6f, \"seconds_shim_record_only\": %.6f, \"simd\": %d, \"isp_64x2\": %d}\n", n_pic, W, H,
               g_decode_seconds_pass[0], g_decode_seconds_pass[1], g_simd, g_isp_64x2);
        return 0;
    }
    if (g_isp_64x2) { fprintf(stderr, "gen_pipe: the stream holds %d 64x2 ISP partitions (reference result undefined): pick another seed\n", g_isp_64x2); return 1; }

    if (!want_shim) {
        char fn[256];
        snprintf(fn, sizeof(fn), "%s.ovg", name);
        gfile g = gfile_open(dir, fn);
        uint32_t d2[2] = { (uint32_t)n_pic, 14 };
        gfile_array(&g, "info", T_I32, out.info.data, 2, d2);
        int32_t geo[4] = { W, H, n_pic, 7 };
        uint32_t d1 = 4;
        gfile_array(&g, "geometry", T_I32, geo, 1, &d1);
        gfile_buf(&g, "frames", &out.frames);
        d2[0] = (uint32_t)(g_dmvr_log.n / 12); d2[1] = 12;
        gfile_array(&g, "dmvr", T_I32, g_dmvr_log.data ? g_dmvr_log.data : (const void *)"", 2, d2);
        gfile_close(&g);
        fprintf(stderr, "%s: %d pictures %dx%d, %u DMVR calls\n", fn, n_pic, W, H, d2[0]);
        return 0;
    }
    if (want_dev) {
        /* the device half recorded what record-only mode recorded */
        for (int i = 0; i < SHIM_NARR; ++i) {
            const size_t bytes = out.S.arr[i].n * g_tsize[out.S.arr[i].type];
            if (out.S.arr[i].n != out.S2.arr[i].n || (bytes && memcmp(out.S.arr[i].data, out.S2.arr[i].data, bytes))) {
                fprintf(stderr, "gen_pipe: device pass: recorder array %d differs from the record-only pass\n", i); return 1;
            }
        }
        char fn[256];
        snprintf(fn, sizeof(fn), "shim_%s_dev.ovg", name);
        gfile g = gfile_open(dir, fn);
        uint32_t d2[2] = { (uint32_t)(g_events.n / sizeof(ovhip_frame_event)), sizeof(ovhip_frame_event) };
        gfile_array(&g, "events", T_U8, g_events.data, 2, d2);
        int32_t geo[4] = { W, H, n_pic, 7 };
        uint32_t d1 = 4;
        gfile_array(&g, "geometry", T_I32, geo, 1, &d1);
        gfile_close(&g);
        fprintf(stderr, "%s: %u events\n", fn, d2[0]);
        return 0;
    }
    /* shim_pipe.ovg: shim_stream_write's container (one case per picture) + the picture-level parameters the flush would take */
    {
        static const char *nm[SHIM_NARR] = { "tb", "coef", "mc", "mcx", "aff", "side", "region", "ciip", "edge_v", "edge_h", "itask" };
        struct shim_stream *S = &out.S;
        char fn[256];
        snprintf(fn, sizeof(fn), "shim_%s.ovg", name);
        gfile g = gfile_open(dir, fn);
        uint32_t end[SHIM_NARR], d2[2];
        for (int i = 0; i < SHIM_NARR; ++i) {
            const size_t es = shim_elem(i) / g_tsize[S->arr[i].type];
            end[i] = (uint32_t)(S->arr[i].n / es);
            d2[0] = end[i]; d2[1] = (uint32_t)es;
            gfile_array(&g, nm[i], S->arr[i].type, S->arr[i].data ? S->arr[i].data : (const void *)"", 2, d2);
        }
        gbuf_push(&S->off, end, SHIM_NARR);
        d2[0] = S->n_cases + 1; d2[1] = SHIM_NARR;
        gfile_array(&g, "case_off", T_U32, S->off.data, 2, d2);
        const uint32_t n_ctb = (uint32_t)(seq.nb_ctb_w * seq.nb_ctb_h);
        uint32_t d3[3] = { (uint32_t)n_pic, n_ctb, sizeof(ovhip_sao_ctu) };
        gfile_array(&g, "sao", T_U8, out.sao.data, 3, d3);
        d3[2] = sizeof(ovhip_alf_ctu);
        gfile_array(&g, "alf_ctus", T_U8, out.alf.data, 3, d3);
        static const char *tn[5] = { "luma_coeff", "luma_clip", "chroma_coeff", "chroma_clip", "cc_coeff" };
        for (int t = 0; t < 5; ++t) { d2[0] = (uint32_t)n_pic; d2[1] = (uint32_t)(out.tab[t].n / n_pic); gfile_array(&g, tn[t], T_I16, out.tab[t].data, 2, d2); }
        d2[0] = (uint32_t)n_pic; d2[1] = sizeof(ovhip_lmcs_luts); gfile_array(&g, "luts", T_U8, out.luts.data, 2, d2);
        d2[1] = sizeof(ovhip_dbf_offsets);                        gfile_array(&g, "dbf_offsets", T_I8, out.offs.data, 2, d2);
        d2[1] = 16;                                               gfile_array(&g, "ref_map", T_I32, out.refmap.data, 2, d2);
        d2[1] = 4;                                                gfile_array(&g, "pic_flags", T_I32, out.pflags.data, 2, d2);
        gfile_close(&g);
        fprintf(stderr, "%s: %u pictures", fn, S->n_cases);
        for (int i = 0; i < SHIM_NARR; ++i) if (end[i]) fprintf(stderr, ", %u %s", end[i], nm[i]);
        fprintf(stderr, "\n");
    }
    return 0;
}
