/* ovvc_hip.h -- C ABI of the MI355X (gfx950) reconstruction back-end for OpenVVC.
 *
 * This is the drop-in boundary: a plain-C shared library (libovvc_hip.so) that a host
 * decoder written in C binds exactly where OpenVVC binds its x86/ARM back-ends -- the
 * `struct RCNFunctions` dispatch table (reference: libovvc/rcn_structures.h:499-694, filled
 * by rcn_init_functions(), libovvc/rcn.c:147-300).  The reference drives reconstruction one
 * block per call, interleaved with CABAC parsing; a GPU cannot be driven that way, so the
 * boundary is split in two layers (INTEGRATION.md shows the reference-side stub):
 *
 *   1. RECORDER  (host, no GPU needed): ovhip_rec_*() take the arguments the reference's
 *      orchestrator slots receive (rcn_tu_st / rcn_mcp_b / ...), run the host-side control
 *      logic those slots contain (transform-type selection, QP -> scale/shift, LFNST kernel
 *      choice, MV clipping, identical-motion test, PU splitting) and append fixed-size
 *      commands to per-picture buffers.
 *   2. ENGINE    (device): ovhip_*_launch() run one frame-resident, stage-parallel HIP kernel
 *      per stage over a whole command buffer (inverse quant + LFNST + inverse transform +
 *      residual add; luma/chroma motion compensation; deblocking; SAO; ALF/CC-ALF).
 *
 * Signatures are plain pointers and sizes; no torch / C++ types.  Every engine entry point
 * returns 0 on success or a negative OVHIP_E* code (the reference slots return void -- see
 * SURVEY.md 8b "Error convention"; the shim latches the first error per picture) and fails
 * loudly (OVHIP_ENODEV) when no HIP device is present: there is no CPU fallback.
 *
 * All sample planes are 16-bit (OVSample = uint16_t for 10-bit, libovvc/bitdepth.h:36-40),
 * 4:2:0, strides in SAMPLES.  10-bit only, like the reference's x86 SIMD path (rcn.c:217).
 */
#ifndef OVVC_HIP_H
#define OVVC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OVHIP_ABI_VERSION 8

/* ---- error codes (negative, in the spirit of libovvc/overror.h:40-45) ---- */
#define OVHIP_OK        0
#define OVHIP_ENODEV   (-1)  /* no HIP device / runtime failure                  */
#define OVHIP_ENOMEM   (-2)
#define OVHIP_EINVAL   (-3)  /* malformed command / argument                      */
#define OVHIP_ELAUNCH  (-4)  /* kernel launch or execution error (see last_error) */
#define OVHIP_EUNSUP   (-5)  /* tool outside the supported hot path (e.g. RPR)    */
#define OVHIP_EREF     (-6)  /* a reference picture failed to decode, or the DPB was shut down while waiting for it */

/* ---- transform types: same numbering as enum DCTType, rcn_structures.h:87-93 ---- */
enum { OVHIP_DST_VII = 0, OVHIP_DCT_VIII = 1, OVHIP_DCT_II = 2 };

/* ------------------------------------------------------------------------------------
 * Picture: three device-resident planes.  Mirrors what the rcn path sees of an OVFrame
 * (libovvc/ovframe.h:84-124: data[3], linesize[3], width, height) -- tight planes, no
 * padding (ovframepool.c set_plane_properties); reference windows that leave the picture
 * are resolved by coordinate clamping on the device instead of emulate_block_border()
 * (rcn_inter.c:148-225).
 * ---------------------------------------------------------------------------------- */
typedef struct ovhip_pic {
    uint16_t *y, *cb, *cr;   /* DEVICE pointers                           */
    int32_t   w, h;          /* luma size in samples (chroma = w/2 x h/2) */
    int32_t   stride_y;      /* samples                                   */
    int32_t   stride_c;      /* samples                                   */
} ovhip_pic;

/* ------------------------------------------------------------------------------------
 * Transform-block command: one inverse-quantisation + (LFNST) + 2-pass inverse transform
 * + residual-add.  Produced by ovhip_rec_tu(); replaces one pass through
 * rcn_residual()/rcn_residual_c()/rcn_res_c()/rcn_jcbcr() + ict.add/ict.ict
 * (rcn_transform_tree.c:415-506, :553-628, :720-867, :1228-1301; rcn_residuals.c:46-222).
 * 32 bytes.
 * ---------------------------------------------------------------------------------- */
enum {                        /* ovhip_tb_cmd.kind */
    OVHIP_TB_TR      = 0,     /* dequant -> [LFNST] -> vertical pass (>>7) -> horizontal (>>10) */
    OVHIP_TB_DC      = 1,     /* DC-only shortcut, inverse_dct_ii_dc (rcn_transform.c:576-598)  */
    OVHIP_TB_TS      = 2,     /* transform skip: de-scan + dequant_ts, no transform             */
    OVHIP_TB_TS_RAW  = 3      /* transform skip, coefficients already raster + scaled (memcpy)  */
};
enum {                        /* ovhip_tb_cmd.res_mode: how the residual r is applied (rcn_residuals.c) */
    OVHIP_RES_ADD      = 0,   /*  r        */
    OVHIP_RES_SUB      = 1,   /* -r        */
    OVHIP_RES_ADD_HALF = 2,   /*  r >> 1   */
    OVHIP_RES_SUB_HALF = 3,   /* (-r) >> 1 */
    OVHIP_RES_SCALE    = 4,   /* flag: LMCS chroma residual scaling with c_scale (scale_* variants) */
    OVHIP_RES_SCALE_IDX = 8,  /* flag (with OVHIP_RES_SCALE): c_scale is an INDEX into the launch's table of
                               * device-derived scales (ovhip_lmcs_scale_launch), not the scale itself */
    OVHIP_RES_STORE = 16      /* the block belongs to an ordered task (ovhip_itask): r (after the sign / half variant,
                               * before any chroma scaling) is STORED into the residual picture of the launch, saturated
                               * to int16, instead of being added -- the ordered pass adds it to its prediction */
};
#define OVHIP_TB_FLAG_RASTER 0x80  /* in .kind: coefficients stored raster (2xN / Nx2 chroma TBs) */
#define OVHIP_TB_FLAG_BDPCM  0x40  /* in .kind (with TS / TS_RAW): block DPCM, rcn_bdpcm_tb (rcn_transform_tree.c:631-688):
                                    * the LEVELS are accumulated along rows (tr_h = 0) or columns (tr_h = 1) with int16
                                    * saturation, THEN de-quantised (TS) or taken as they are (TS_RAW)               */

typedef struct ovhip_tb_cmd {
    uint16_t x, y;            /* top-left in samples of `plane`                                 */
    uint8_t  plane;           /* 0 Y, 1 Cb, 2 Cr                                                */
    uint8_t  log2_w, log2_h;  /* 1..6                                                           */
    uint8_t  kind;            /* OVHIP_TB_* | OVHIP_TB_FLAG_RASTER                              */
    uint8_t  tr_h, tr_v;      /* OVHIP_DST_VII / DCT_VIII / DCT_II                              */
    uint8_t  lfnst;           /* bit0 on, bits1-2 kernel set (lfnst_mode_map), bit3 idx, bit4 transpose */
    uint8_t  res_mode;        /* OVHIP_RES_*                                                    */
    uint8_t  plane2;          /* 0xff none; else second destination (JCCR), same x,y            */
    uint8_t  res_mode2;
    uint8_t  dq_shift;        /* |shift| of struct IQScale (rcn_dequant.c:92-158)                */
    uint8_t  dq_neg;          /* 1: c * (scale << shift)   0: (c*scale + add) >> shift          */
    int16_t  dq_scale;
    int16_t  c_scale;         /* LMCS chroma residual scale (1<<11 = none)                      */
    uint32_t coef_off;        /* int16 index into the coefficient arena                         */
    uint64_t sig_sb_map;      /* bit sb_y*8+sb_x; arena holds 16 int16 per SET bit, ascending   */
} ovhip_tb_cmd;

/* ------------------------------------------------------------------------------------
 * Prediction-unit command ("MC unit"): <= 16x16 luma samples (+ the co-located 4:2:0 chroma)
 * predicted from one or two reference pictures.  Produced by ovhip_rec_pu(), which performs
 * clip_mv() (rcn_inter.c:96-109), the identical-motion test (:256-268), half-pel AMVR filter
 * selection (:572-577) and BCW weight lookup (:89, :587-596) on the host, then splits the PU.
 * Replaces rcn_mcp / rcn_mcp_b / rcn_mcp_b_l / rcn_mcp_b_c and the leaf kernels behind
 * mc_l/mc_c.{unidir,bidir0,bidir1,bidir_w} (rcn_inter.c:520-602, :1391-1554, :2750-2966;
 * rcn_mc.c:382-1610).  32 bytes.
 * ---------------------------------------------------------------------------------- */
enum {                        /* ovhip_mc_unit.flags */
    OVHIP_MC_HPEL_FILT = 1,   /* prec_amvr == MV_PRECISION_HALF: frac 8 uses the 6-tap smoothing row */
    OVHIP_MC_FILT_4x4  = 2,   /* the reference call was a 4x4 luma block: ov_mc_filters_4        */
    OVHIP_MC_NO_LUMA   = 4,   /* rcn_mcp_b_c: chroma only                                        */
    OVHIP_MC_NO_CHROMA = 8,   /* rcn_mcp_b_l: luma only                                          */
    OVHIP_MC_LMCS      = 16,  /* forward-reshape the luma prediction (lmcs_reshape_forward)      */
    /* "refined" units: only accepted by ovhip_mcx_launch(); dir == 3, plain average, w,h in {8,16}     */
    OVHIP_MC_BDOF      = 32,  /* luma through bi-directional optical flow: rcn_bdof_mcp_l
                               * (rcn_inter.c:1136-1250; rcn_prof_bdof.c:303-490)                  */
    OVHIP_MC_GPM       = 128, /* geometric partitioning (rcn_gpm_b, rcn_inter.c:3118-3143): dir == 3, ref0/mv0 and
                               * ref1/mv1 are the two uni-predictions, blended with the per-sample weight
                               * w = clip3(0, 8, (K + A*x + B*y) >> 3): (p0*w + p1*(8-w) + 64) >> 7,
                               * put_weighted_gpm_bi_pixels (rcn_mc.c:1630-1655); aux = (K & 0xffff) | (A & 0xff) << 16 | (B & 0xff) << 24,
                               * x, y relative to the unit (chroma samples use 2x, 2y)             */
    OVHIP_MC_DMVR      = 64   /* decoder-side MV refinement first: rcn_dmvr_mv_refine
                               * (rcn_inter.c:872-1126).  mv0/mv1 are then NOT clipped (the device
                               * applies clip_mv for the window anchor only, as the reference does);
                               * with OVHIP_MC_BDOF = its apply_bdof argument                      */
};

typedef struct ovhip_mc_unit {
    uint16_t x, y;            /* luma position in the picture                                    */
    uint8_t  w, h;            /* luma size, 4..16 (chroma w/2 x h/2)                             */
    uint8_t  dir;             /* 1: ref0 only, 2: ref1 only, 3: bi                               */
    uint8_t  flags;           /* OVHIP_MC_*                                                      */
    uint8_t  ref0, ref1;      /* indices into the launch's reference-picture table               */
    int8_t   w0, w1;          /* bi weights (4,4 = plain average path; else BCW, w0+w1 == 8)     */
    int32_t  mv0x, mv0y;      /* 1/16 luma pel, ALREADY clipped (chroma uses the same at 1/32)   */
    int32_t  mv1x, mv1y;
    uint32_t aux;             /* OVHIP_MC_GPM: packed weight plane (see the flag).  Otherwise 0, or the fused CIIP
                               * blend: bits 0-2 wt (1..3), bit 8 = chroma keeps the inter prediction            */
} ovhip_mc_unit;

/* ------------------------------------------------------------------------------------
 * Affine unit: <= 16x16 luma samples of an affine CU, every 4x4 luma sub-block with its own
 * motion vectors and (optionally) prediction refinement with optical flow (PROF); the chroma of
 * the same area as 4x4-chroma blocks with the averaged motion vector.  Produced by
 * ovhip_rec_affine_cu().  Replaces the per-sub-block calls of rcn_affine_mcp_b_l /
 * rcn_affine_prof_mcp_b_l / rcn_affine_mcp_b_c (drv_affine_mvp.c:3264-3411) into
 * rcn_mcp_b_l(2,2) / rcn_prof_mcp_b_l / rcn_mcp_b_c(3,3) and the leaf code behind them:
 * rcn_prof_motion_compensation_b_l, rcn_prof_mcp_l (rcn_inter.c:1252-1386, :1631-1722),
 * extend_prof_buff / compute_prof_grad / rcn_prof (rcn_prof_bdof.c:152-290).  32 bytes.
 *
 * Side arena (int32): at side_off, for each luma sub-block in raster order 4 words
 * {mv0x, mv0y, mv1x, mv1y} (clip_mv() already applied for a 4x4 block), then for each
 * 8x8-luma chroma block 4 words (clip_mv() for the 8x8 block).  At prof_off (CU-wide, only
 * read when OVHIP_AFF_PROF is set): struct PROFInfo = 64 int16 {h0[16], v0[16], h1[16], v1[16]}.
 * ---------------------------------------------------------------------------------- */
enum {                        /* ovhip_aff_unit.flags */
    OVHIP_AFF_PROF      = 1,  /* the CU went through rcn_affine_prof_mcp_b_l                      */
    OVHIP_AFF_LMCS      = 16, /* = OVHIP_MC_LMCS                                                  */
    OVHIP_AFF_NO_CHROMA = 8   /* = OVHIP_MC_NO_CHROMA                                             */
};

typedef struct ovhip_aff_unit {
    uint16_t x, y;            /* luma position in the picture                                    */
    uint8_t  w, h;            /* luma size, multiples of 8, <= 16                                 */
    uint8_t  dir;             /* 1, 2, 3                                                         */
    uint8_t  flags;           /* OVHIP_AFF_*                                                     */
    uint8_t  ref0, ref1;
    int8_t   w0, w1;          /* bi weights as in ovhip_mc_unit                                  */
    uint8_t  prof_dir;        /* bi-prediction: lists refined by PROF (bit0 L0, bit1 L1)         */
    uint8_t  ident_c;         /* per chroma block: identical motion -> uni-prediction from L1    */
    uint16_t ident_l;         /* per luma sub-block (no PROF only): same, rcn_inter.c:256-268    */
    uint32_t side_off;        /* int32 index of the unit's motion vectors in the side arena      */
    uint32_t prof_off;        /* int32 index of the CU's PROFInfo in the side arena              */
    uint32_t pad[2];
} ovhip_aff_unit;

/* ------------------------------------------------------------------------------------
 * CIIP blend unit: dst = (intra * wt + inter * (4 - wt) + 2) >> 2 over one CU, luma and chroma
 * (put_weighted_ciip_pixels rcn_mc.c:1611-1628 driven by rcn_ciip_weighted_sum rcn_inter.c:2968-3009).
 * The inter prediction is what the MC launch left in `dst` (rcn_ciip / rcn_ciip_b record the PU
 * as usual); the intra (planar) prediction comes from the caller's intra path in a second picture.
 * 8 bytes.
 * ---------------------------------------------------------------------------------- */
typedef struct ovhip_ciip_unit {
    uint16_t x, y;            /* luma position in the picture                                    */
    uint8_t  log2_w, log2_h;  /* CU size                                                         */
    uint8_t  wt;              /* 1 + (above CU intra) + (left CU intra)                           */
    uint8_t  chroma_inter;    /* log2_w <= 2: chroma keeps the inter prediction (rcn_inter.c:2998)*/
} ovhip_ciip_unit;

/* ------------------------------------------------------------------------------------
 * LMCS (luma mapping with chroma scaling), rcn_lmcs.c.
 *   ovhip_lmcs_build()      = rcn_init_lmcs -> init_lmcs_lut (rcn_lmcs.c:93-179, :352-370): host, per APS
 *   forward reshape          : fused into the MC kernels (OVHIP_MC_LMCS + fwd LUT)
 *   ovhip_lmcs_scale_launch = rcn_lmcs_compute_chroma_scale (rcn_lmcs.c:204-350) for every 64-aligned CU
 *                             (vcl_coding_unit.c:724-730), on the reshaped-domain reconstruction
 *   ovhip_lmcs_inverse_launch = lmcs_reshape_backward over the picture (slicedec.c:746-750, :799-803)
 * ---------------------------------------------------------------------------------- */
typedef struct ovhip_lmcs_data {       /* struct OVLMCSData (nvcl_structures.h:671-680) with the signs applied */
    uint8_t  min_bin_idx;              /* lmcs_min_bin_idx                                                  */
    uint8_t  delta_max_bin_idx;        /* lmcs_delta_max_bin_idx                                            */
    int16_t  crs_offset;               /* +-lmcs_delta_abs_crs                                              */
    int16_t  cw_delta[16];             /* +-lmcs_delta_abs_cw[i]                                            */
} ovhip_lmcs_data;

typedef struct ovhip_lmcs_luts {       /* struct LMCSLUTs + the LMCSInfo scalars (rcn_lmcs.c:75-81, rcn_lmcs.h:40-49) */
    uint16_t fwd_lut[1024];
    uint16_t bwd_lut[1024];
    uint16_t wnd_bnd[17];
    uint8_t  min_idx, max_idx;
    int16_t  crs_offset;
    uint16_t pad;
} ovhip_lmcs_luts;

typedef struct ovhip_lmcs_region {     /* one rcn_lmcs_compute_chroma_scale() call, 8 bytes */
    uint16_t x, y;                     /* luma position of the 64-aligned CU in the picture                 */
    uint8_t  n_abv, n_lft;             /* available 4-sample units above / left (bit length of the masks)    */
    uint8_t  ordered;                  /* 1: luma around the region comes from ordered tasks -- ovhip_lmcs_scale_launch skips
                                        * it, an OVHIP_IT_REGION task derives the scale in the ordered pass  */
    uint8_t  pad;
} ovhip_lmcs_region;

int ovhip_lmcs_build(const ovhip_lmcs_data *data, ovhip_lmcs_luts *out);


/* ------------------------------------------------------------------------------------
 * Deblocking.  The reference filters one CTU per df.rcn_dbf_ctu() call from CTU-local bit maps
 * (struct DBFInfo, libovvc/ctudec.h:130-170) that carry neighbour state between calls
 * (dbf_load_info/dbf_store_info, drv_lines.c:618-761).  The recorder (ovhip_rec_dbf_ctu) runs the
 * edge enumeration, bS / QP lookup and filter-length derivation of vvc_dbf_ctu_hor/_ver and
 * vvc_dbf_chroma_hor/_ver (rcn_df.c:1151-1431, :1876-2167) WITHOUT touching samples and writes one
 * 16-bit parameter word per 4-sample edge segment into picture-level planes; the device then
 * filters all vertical edges of the picture, then all horizontal edges (ovhip_dbf_launch), which
 * yields the same samples as the reference's per-CTU V-then-H order.
 *
 * luma word   : bits 0-1 bS (0 = edge not filtered), 2-4 max filter length P side (1,2,3,5,7),
 *               5-7 max filter length Q side, 8-15 average QP byte of the two blocks
 *               ((qp_p + qp_q + 1) >> 1 on the uint8 map entries, as the reference computes it).
 * chroma word : bit 0 filtered, bit 1 bS==2, bit 2 "large" (strong filter allowed), bit 3 CTU-top
 *               horizontal edge (is_ctb_b: one P-side line), bits 8-15 average chroma QP byte.
 * Planes are indexed by 4x4-luma-sample unit: luma_v[uy*w4 + ux] = vertical edge at x = 4*ux, rows
 * 4*uy..4*uy+3; luma_h[uy*w4 + ux] = horizontal edge at y = 4*uy, columns 4*ux..+3.  Chroma edges
 * live on the 8-luma-sample grid: c*_v[uy*(w4/2) + ux/2], c*_h[(uy/2)*w4 + ux] (2 chroma samples).
 * ---------------------------------------------------------------------------------- */
#define OVHIP_DBF_LUMA(bs, lp, lq, qp)  ((uint16_t)((bs) | ((lp) << 2) | ((lq) << 5) | ((qp) << 8)))
#define OVHIP_DBF_C_ON     1
#define OVHIP_DBF_C_BS2    2
#define OVHIP_DBF_C_LARGE  4
#define OVHIP_DBF_C_CTB_B  8

typedef struct ovhip_dbf_planes {
    const uint16_t *luma_v, *luma_h;      /* [h4][w4]                      */
    const uint16_t *cb_v, *cr_v;          /* [h4][w4c],  w4c = (w4+1)/2    */
    const uint16_t *cb_h, *cr_h;          /* [h4c][w4],  h4c = (h4+1)/2    */
    int32_t w4, h4;
    int16_t beta_offset, tc_offset;       /* DBFInfo.beta_offset / tc_offset (per slice) */
} ovhip_dbf_planes;

/* What df.rcn_dbf_ctu / df.rcn_dbf_truncated_ctu receive (rcn_structures.h:408-413): a copy of the
 * CTU's struct DBFInfo arrays (same element layout: 16+33 / 33 x uint64 masks, 34x33 QP bytes), after
 * ovhip_rec_dbf_mv_prepass() in P / B slices, plus the call arguments and ctudec->ctu_ngh_flags. */
typedef struct ovhip_dbf_ctu {
    uint64_t ctb_bound_ver[49], ctb_bound_hor[49], ctb_bound_ver_c[49], ctb_bound_hor_c[49];
    uint64_t aff_edg_ver[49], aff_edg_hor[49];
    uint64_t bs2_ver[33], bs2_hor[33], bs2c_ver[33], bs2c_hor[33];
    uint64_t bs1_ver[33], bs1_hor[33], bs1cb_ver[33], bs1cb_hor[33], bs1cr_ver[33], bs1cr_hor[33];
    uint64_t affine_ver[33], affine_hor[33];
    uint8_t  qp_y[34 * 33], qp_cb[34 * 33], qp_cr[34 * 33];
    int16_t  beta_offset, tc_offset;
    uint8_t  disable_v, disable_h;
    uint8_t  log2_ctu_s, last_x, last_y;  /* slot arguments                                    */
    uint8_t  ctu_lft, ctu_abv;            /* ctu_ngh_flags & CTU_LFT_FLG / CTU_UP_FLG           */
    uint8_t  pad;
    uint16_t ctu_w, ctu_h;                /* samples; < 1<<log2_ctu_s selects the truncated slot */
    uint16_t ctb_x, ctb_y;                /* CTU address in the picture                         */
} ovhip_dbf_ctu;

/* The MV-based boundary-strength pre-pass rcn_dbf_ctu runs first in P / B slices (dbf_ctu_preproc_v/_h ->
 * dbf_mv_set_vedges/_hedges -> check_dbf_enabled(_p), mv_threshold_check; rcn_df.c:1512-1874): CU and affine
 * sub-block edges that carry no bS yet get bS 1 when the motion on the two sides differs (different reference
 * pictures, or a vector component >= 8 in 1/16 units apart).  Host work on the CTU's 34x34 motion grids. */
typedef struct ovhip_dbf_mv_ctx {
    uint64_t cu_edge_ver[33], cu_edge_hor[33];           /* dbf_info->cu_edge (dbf_utils.h:60-73)                     */
    uint64_t map0_h[33], map0_v[33];                     /* inter_ctx->mv_ctx0.map.hfield / vfield (ctudec.h:298-302)  */
    uint64_t map1_h[33], map1_v[33];                     /* inter_ctx->mv_ctx1.map                                    */
    uint64_t ibc_h[33], ibc_v[33];                       /* dbf_info->ibc_ctx->ctu_map (all 0 without IBC)             */
    int16_t  dist_ref0[16], dist_ref1[16];               /* inter_ctx->dist_ref_0 / _1                                */
    const void *mvs0, *mvs1;                             /* inter_ctx->mv_ctx0.mvs / mv_ctx1.mvs: OVMV[34 * 34]        */
    int32_t  mv_bytes;                                   /* sizeof(OVMV): x, y int32 at offset 0 / 4, ref_idx int8 at 8 */
} ovhip_dbf_mv_ctx;
/* Updates ctu->bs1_ver / bs1_hor in place (what the slot does to dbf_info->bs1_map); call it before
 * ovhip_rec_dbf_ctu for slices other than I. */
int ovhip_rec_dbf_mv_prepass(ovhip_dbf_ctu *ctu, const ovhip_dbf_mv_ctx *mv);

/* ------------------------------------------------------------------------------------
 * Sample adaptive offset.  One entry per CTU in raster order (index ctb_y * nb_ctu_w + ctb_x,
 * nb_ctu_w = ceil(w / ctu)), a compact copy of what the SAO slots read from SAOParamsCtu
 * (libovvc/dec_structures.h:263-274: type_idx, band_position, eo_class, offset_val).
 * ovhip_sao_launch() reads the deblocked picture `src` and writes every sample of `dst` (copy
 * where SAO is off): the frame-resident equivalent of the reference's filter_region copies
 * (rcn_ctu.c:315-510) -- a sample is always filtered with the parameters of the CTU that
 * contains it and picture-border samples whose neighbour is outside stay unmodified
 * (rcn_sao.c:119-188, :190-293).
 * ---------------------------------------------------------------------------------- */
enum { OVHIP_SAO_OFF = 0, OVHIP_SAO_BAND = 1, OVHIP_SAO_EDGE = 2 };   /* = SAO_NOT_APPLIED / SAO_BAND / SAO_EDGE */
/* Sides of a CTU that are borders of its RECT ENTRY (tile) inside the picture: the reference filters a rect entry on its own
 * (slicedec.c:636-657) -- is_border comes from the entry-local CTU index (rcn_sao.c:211-214, :253-257; rcn_alf.c:1313-1318): SAO
 * leaves a sample whose neighbour lies across such a side unmodified, ALF / CC-ALF pad across it as at the picture border
 * (rcn_extend_filter_region, rcn_ctu.c:361-508).  All zero in a picture of one entry.  ONE_ROW: the entry is a single CTU row
 * high (its first 6-row band is then filtered with the BOTTOM flag set, rcn_sao.c:262). */
#define OVHIP_BORDER_LEFT    1
#define OVHIP_BORDER_RIGHT   2
#define OVHIP_BORDER_UPPER   4
#define OVHIP_BORDER_BOTTOM  8
#define OVHIP_BORDER_ONE_ROW 16
typedef struct ovhip_sao_ctu {
    uint8_t type[3];          /* per component Y, Cb, Cr                                  */
    uint8_t band_position[3]; /* first of the 4 consecutive bands (of 32)                  */
    uint8_t eo_class[3];      /* 0 horizontal, 1 vertical, 2 45 deg (135 in spec), 3 other diagonal */
    uint8_t border;           /* OVHIP_BORDER_*                                            */
    uint8_t pad[2];
    int16_t offset_val[3][5]; /* band: [0..3]; edge: indexed by 2 + sign(c-a) + sign(c-b) */
    uint8_t pad2[2];
} ovhip_sao_ctu;

/* ------------------------------------------------------------------------------------
 * Adaptive loop filter + cross-component ALF.  Per CTU: what alf.rcn_alf_filter_line reads from
 * ALFParamsCtu (dec_structures.h:277-283) and alf_info.ctb_cc_alf_filter_idx (ctudec.h:184-215).
 * Per picture: the coefficient / clip sets exactly as rcn_alf_reconstruct_coeff_APS() expands them
 * on the host (struct RCNALF, rcn_alf.h:60-66: 16 fixed + 8 APS luma sets of 4 transposes x 25
 * classes x 13 taps; 8 chroma alternatives x 7) and the CC-ALF coefficients of the two APS
 * (OVALFData.alf_cc_mapped_coeff[2][4][8], nvcl_structures.h:668).  ovhip_alf_launch() reads the
 * post-SAO picture `src` (clamped at the picture border = the replicate-padded filter_region,
 * rcn_ctu.c:361-508) and writes every sample of `dst`.
 * ---------------------------------------------------------------------------------- */
#define OVHIP_ALF_LUMA_SET_SIZE (4 * 25 * 13)
typedef struct ovhip_alf_ctu {
    uint8_t flags;            /* ctb_alf_flag: bit2 luma, bit1 Cb, bit0 Cr                        */
    uint8_t luma_set;         /* ctb_alf_idx: 0..15 fixed sets, 16.. APS sets                     */
    uint8_t cb_alt, cr_alt;   /* chroma alternative filter index                                  */
    uint8_t cc_cb_idx, cc_cr_idx; /* ctb_cc_alf_filter_idx (0 = off, else filter idx + 1)         */
    uint8_t border;           /* OVHIP_BORDER_* (rect-entry borders inside the picture)            */
    uint8_t pad;
} ovhip_alf_ctu;

typedef struct ovhip_alf_pic {
    const ovhip_alf_ctu *ctus;    /* DEVICE [ceil(w/ctu) * ceil(h/ctu)], raster                    */
    const int16_t *luma_coeff;    /* DEVICE [24][OVHIP_ALF_LUMA_SET_SIZE]  RCNALF.filter_coeff_dec  */
    const int16_t *luma_clip;     /* DEVICE [24][OVHIP_ALF_LUMA_SET_SIZE]  RCNALF.filter_clip_dec   */
    const int16_t *chroma_coeff;  /* DEVICE [8][7]  RCNALF.chroma_coeff_final                      */
    const int16_t *chroma_clip;   /* DEVICE [8][7]  RCNALF.chroma_clip_final                       */
    const int16_t *cc_coeff;      /* DEVICE [2][4][8]  Cb APS / Cr APS alf_cc_mapped_coeff[c]      */
    uint8_t *class_scratch;       /* DEVICE scratch, ceil(w/4) * ceil(h/4) bytes (class | transpose << 5) */
    int32_t log2_ctu_s;
} ovhip_alf_pic;

/* ------------------------------------------------------------------------------------
 * Ordered tasks: everything whose INPUT is reconstructed samples of the same picture -- intra prediction
 * (vvc_intra_pred, vvc_intra_pred_chroma, intra_pred_mrl, rcn_intra_mip, cclm.*: rcn_structures.h:507-524, :558-593;
 * rcn_intra.c:484-1180, rcn_intra_cclm.c, rcn_intra_mip.c, rcn_fill_ref.c), CIIP's planar part (rcn_inter.c:3011-3067), and
 * the chroma-scale regions / chroma residuals that depend on such blocks.  The reference runs them in decoding order,
 * interleaved with everything else; the recorder gives each task a LEVEL = 1 + the highest level among the tasks that
 * produce its inputs (0 = inputs come from inter blocks only), and the device runs level after level, all tasks of a level
 * in one launch, after the stage-parallel prediction / residual launches.  One task = one prediction block (one
 * transform block of an intra CU): predict, add the residual the transform stage STOREd for it, clip, write.  32 bytes.
 * ---------------------------------------------------------------------------------- */
enum { OVHIP_IT_LUMA = 0,     /* luma block                                                                          */
       OVHIP_IT_CHROMA = 1,   /* Cb + Cr block (x, y, size in chroma samples)                                        */
       OVHIP_IT_REGION = 2,   /* rcn_lmcs_compute_chroma_scale of region c_scale (needs ordered luma around it)      */
       OVHIP_IT_RES_C = 3 };  /* chroma residual add of an already predicted block whose scale is an ordered region's */
enum {                        /* ovhip_itask.flags */
    OVHIP_IF_CORNER = 1,      /* the above-left neighbour unit is available                                          */
    OVHIP_IF_MIP = 2,         /* matrix-based intra prediction, mode = mip mode; OVHIP_IF_MIP_TR: transposed          */
    OVHIP_IF_MIP_TR = 4,
    OVHIP_IF_BDPCM = 8,       /* block-DPCM CU: pure horizontal copy, OVHIP_IF_BDPCM_VER: vertical                    */
    OVHIP_IF_BDPCM_VER = 16,
    OVHIP_IF_RES_Y = 32, OVHIP_IF_RES_CB = 64, OVHIP_IF_RES_CR = 128,   /* a STOREd residual exists for that plane    */
    OVHIP_IF_RES_SCALE = 256, /* chroma residual is LMCS-scaled with c_scale ...                                     */
    OVHIP_IF_SCALE_IDX = 512, /* ... which is the index of a chroma-scale region                                      */
    OVHIP_IF_CORNER_L = 2048, /* ISP only: the corner unit as the LEFT arm's progress map sees it (OVHIP_IF_CORNER: as the above
                               * arm's map sees it; for every other task the two coincide)                                */
    OVHIP_IF_ISP = 1024       /* a prediction call of an intra-sub-partition CU (intra_pred_isp, rcn_intra.c:566-640): the
                               * reference arms are the CODING UNIT's (isp_* fields), shifted by the partition's offset; always
                               * the 4-tap cubic filter, no reference smoothing, wide angles by the CU's shape, PDPC only for
                               * blocks at least 4 samples high                                                           */
};
typedef struct ovhip_itask {
    uint16_t x, y;            /* top-left in samples of the task's plane(s), picture coordinates                      */
    uint8_t  log2_w, log2_h;
    uint8_t  kind;            /* OVHIP_IT_*                                                                           */
    uint8_t  mode;            /* 0 planar, 1 DC, 2..66 angular (before the wide-angle remap); chroma also 67 LM, 68 MDLM
                               * left, 69 MDLM top; OVHIP_IF_MIP: the MIP mode                                        */
    uint16_t flags;           /* OVHIP_IF_*                                                                           */
    uint8_t  avl_lft, avl_abv;/* available neighbour units (4 luma / 2 chroma samples) below / right of the corner, as the
                               * reference counts them: position of the highest available unit (rcn_fill_ref.c: 64 -
                               * clz(avl_map) - 1).  LM modes: 67: 0/1 flags; 68: avl_lft = contiguous units, avl_abv flag;
                               * 69: avl_abv = contiguous units, avl_lft flag (rcn_intra_cclm.c:56-68, :770-776, :843-849) */
    uint8_t  mrl_idx;         /* multi-reference-line index 0..2                                                      */
    uint8_t  ciip_wt;         /* != 0: CIIP CU, the (planar) prediction is blended into the inter prediction with this
                               * weight before the residual (ovhip_ciip_weight)                                        */
    int16_t  c_scale;         /* chroma residual scale, or region index (OVHIP_IF_SCALE_IDX, OVHIP_IT_REGION)          */
    uint16_t level;           /* >= 1                                                                                 */
    uint16_t ctu_deps;        /* set by the recorder: 0x8000 | log2_ctu << 8 | neighbour CTUs (bit 0 left, 1 above-left, 2 above,
                               * 3 above-right) holding ordered tasks whose samples this task reads; 0 = unknown            */
    uint8_t  isp_log2_cb_w, isp_log2_cb_h;   /* OVHIP_IF_ISP: the coding unit's size ...                                       */
    uint8_t  isp_off_x, isp_off_y;           /* ... and this block's offset inside it; avl_abv / avl_lft / the corner flag then
                                              * describe the CU's arms (above arm from (x - off_x - 1, y - 1), 2 cb_w + 1 long;
                                              * left arm from (x - 1, y - off_y - 1), 2 cb_h + 1 long)                         */
    uint8_t  isp_log2_pb;     /* OVHIP_IF_ISP: width of one partition inside this block (vertical partitions narrower than 4 are
                               * predicted 4 columns at a time) and ...                                                  */
    uint8_t  isp_res_mask;    /* ... which of them carry a residual (bit = x >> isp_log2_pb); the others add nothing        */
    uint16_t pad[3];
} ovhip_itask;

/* ------------------------------------------------------------------------------------
 * Recorder (host side, pure C, usable without a GPU).
 * ---------------------------------------------------------------------------------- */
typedef struct ovhip_recorder ovhip_recorder;

/* Per-slice / per-CU state the reference keeps in OVCTUDec and that rcn_tu_st & co. read
 * implicitly.  The shim snapshots it at slot-call time. */
typedef struct ovhip_tu_state {
    uint8_t qp_y, qp_cb, qp_cr, qp_jcbcr;                 /* ctudec->dequant_*.qp               */
    uint8_t qp_y_skip, qp_cb_skip, qp_cr_skip, qp_jcbcr_skip;
    uint8_t dep_quant;        /* residual_coding_l == residual_coding_dpq (rcn_transform_tree.c:399) */
    uint8_t mts_implicit;     /* ctudec->mts_implicit (:435)                                      */
    uint8_t sh_ts_disabled;   /* ctudec->sh_ts_disabled (:673)                                    */
    uint8_t ict_type;         /* rcn_init_functions(ict_type): selects ict.ict[][] set (rcn_residuals.c:231) */
    uint8_t lmcs_scale_c;     /* lmcs_info.scale_c_flag: 1 = lmcs_chroma_scale below is the value;
                               * 2 = use the scale of the last ovhip_rec_lmcs_region() (derived on the device) */
    uint8_t pad[3];
    int16_t lmcs_chroma_scale;/* lmcs_info.lmcs_chroma_scale                                      */
    int8_t  intra_mode;       /* ctudec->intra_mode (luma LFNST kernel choice, :461)              */
    int8_t  lfnst_mode_c;     /* chroma: intra mode after derive_lfnst_mode_c's DM/LM substitution */
} ovhip_tu_state;

/* One transform unit as handed to tmp.rcn_tu_st / rcn_tu_l / rcn_tu_c (rcn_structures.h:475-486)
 * together with its struct TUInfo (rcn_transform_tree.c:56-66). */
typedef struct ovhip_tu_desc {
    uint16_t x0, y0;          /* LUMA sample position in the picture (ctb origin already added)  */
    uint8_t  log2_tb_w, log2_tb_h;  /* luma TU size (tree 2: chroma TU size, x0/y0 chroma units)  */
    uint8_t  tree;            /* 0 single tree (rcn_tu_st), 1 luma (rcn_tu_l), 2 chroma (rcn_tu_c) */
    uint8_t  cbf_mask;        /* 0x10 luma, 0x08 joint CbCr, 0x02 Cb, 0x01 Cr                     */
    uint16_t cu_flags;        /* CUFlags bits (cu_utils.h:44-60)                                  */
    uint8_t  tr_skip_mask;    /* 0x10 luma, 0x02 cb, 0x01 cr                                      */
    uint8_t  cu_mts_flag, cu_mts_idx;
    uint8_t  lfnst_flag, lfnst_idx;
    uint8_t  pad;
    uint16_t last_pos[3];     /* tb_info[0]=Cb/joint, [1]=Cr, [2]=luma   (y<<8 | x)               */
    uint64_t sig_sb_map[3];
    const int16_t *coef[3];   /* host pointers: residual_cb/cr/y + pos_offset, reference layout   */
} ovhip_tu_desc;

/* One coding unit's transform tree as handed to tmp.rcn_transform_tree (rcn_structures.h:464-468;
 * rcn_transform_tree.c:1454-1506): the CU is cut at the maximum transform size (and 128-wide CUs at 64 first) and
 * every leaf goes to rcn_tu_st / rcn_tu_l / rcn_tu_c with its own struct TUInfo out of the caller's array of up to
 * 16.  The reference's walker calls those leaves directly (not through the table), so this is the entry a shim
 * has to override; intra CUs additionally need rcn_intra_tu before every leaf, which is the caller's business. */
typedef struct ovhip_tu_info {   /* struct TUInfo (rcn_transform_tree.c:56-66) without the SBT flag */
    uint8_t  cbf_mask, tr_skip_mask, cu_mts_flag, cu_mts_idx, lfnst_flag, lfnst_idx;
    uint16_t pos_offset;          /* int16 offset of this TU's coefficients in ctudec->residual_y / _cb / _cr */
    uint16_t last_pos[3];         /* tb_info[0] = Cb / joint, [1] = Cr, [2] = luma                             */
    uint16_t pad;
    uint64_t sig_sb_map[3];
} ovhip_tu_info;

typedef struct ovhip_tt_desc {
    uint16_t x0, y0;              /* luma position of the CU in the picture (tree 2: chroma units)             */
    uint8_t  log2_w, log2_h;      /* CU size (tree 2: chroma size)                                             */
    uint8_t  log2_max_tb_s;       /* part_ctx->log2_max_tb_s                                                   */
    uint8_t  tree;                /* ctu_dec->transform_unit: 0 transform_unit_st, 1 _l, 2 _c                  */
    uint16_t cu_flags;
    uint16_t pad;
    const ovhip_tu_info *tu_info; /* the caller's array, indexed as the reference does                          */
    const int16_t *residual[3];   /* ctudec->residual_cb, residual_cr, residual_y (bases, pos_offset not applied) */
} ovhip_tt_desc;

/* One prediction call as handed to rcn_mcp_b (rcn_structures.h:640-646). */
typedef struct ovhip_pu_desc {
    uint16_t x0, y0;          /* luma position in the picture                                    */
    uint8_t  log2_w, log2_h;  /* PU size                                                          */
    uint8_t  inter_dir;       /* 1, 2, 3                                                          */
    uint8_t  ref_idx0, ref_idx1;
    uint8_t  bcw_idx_plus1;   /* OVMV.bcw_idx_plus1 of mv0                                        */
    uint8_t  prec_amvr_half;  /* inter_ctx->prec_amvr == MV_PRECISION_HALF                        */
    uint8_t  planes;          /* 3 both (rcn_mcp_b), 1 luma only (rcn_mcp_b_l), 2 chroma only     */
    uint8_t  lmcs;            /* luma forward reshaping active                                    */
    uint8_t  refine;          /* 0: rcn_mcp_b / _l / _c.  bit0 (OVHIP_PU_BDOF): the caller's bdof_enable,
                               * bit1 (OVHIP_PU_DMVR): its dmvr_enable -- the PU is then the whole CU and is
                               * cut into <=16x16 calls exactly as vcl_coding_unit.c:2450-2472 / :2598-2668 do */
    int32_t  mv0x, mv0y, mv1x, mv1y;
    int32_t  poc0, poc1;      /* rpl0[ref_idx0]->poc, rpl1[ref_idx1]->poc (identical-motion test) */
    uint8_t  ref0, ref1;      /* slots of those pictures in the launch's reference table          */
    uint8_t  gpm_split_dir;   /* OVHIP_PU_GPM: gpm_ctx->split_dir (merge_gpm_partition_idx), 0..63  */
    uint8_t  ciip_wt;         /* 0: not a CIIP CU; 1..3 = ovhip_ciip_weight(): the units of this PU blend the
                               * caller's planar prediction in as they are stored (rcn_ciip / rcn_ciip_b),
                               * no separate ovhip_rec_ciip / ovhip_ciip_launch needed                 */
} ovhip_pu_desc;

/* One affine CU as rcn_affine_mcp_b_l / rcn_affine_prof_mcp_b_l / rcn_affine_mcp_b_c receive it
 * (drv_affine_mvp.c:3264-3411): the sub-block motion field the driver wrote into
 * inter_ctx->mv_ctx0/1 and, when prof_dir != 0, the PROFInfo compute_prof_dmv_scale() derived. */
typedef struct ovhip_affine_desc {
    uint16_t x0, y0;          /* luma position in the picture                                    */
    uint8_t  log2_w, log2_h;  /* CU size, >= 8x8                                                 */
    uint8_t  inter_dir;
    uint8_t  bcw_idx_plus1;
    uint8_t  prof_dir;        /* 0: rcn_affine_mcp_b_l, else rcn_affine_prof_mcp_b_l(prof_dir)    */
    uint8_t  lmcs;
    uint8_t  ref0, ref1;      /* slots in the launch's reference table                           */
    int32_t  poc0, poc1;
    int32_t  mv_stride;       /* in motion vectors (34 in the reference's mv_ctx)                 */
    const int32_t *mv0, *mv1; /* host: (x, y) pairs per 4x4 sub-block, row stride mv_stride      */
    int16_t  dmv_scale[4][16];/* struct PROFInfo: h0, v0, h1, v1                                  */
} ovhip_affine_desc;

#define OVHIP_PU_BDOF 1
#define OVHIP_PU_DMVR 2
#define OVHIP_PU_GPM  4       /* rcn_gpm_b: mv0/ref0 and mv1/ref1 are gpm_ctx->mv0 / mv1 and their pictures */

ovhip_recorder *ovhip_rec_create(int32_t pic_w, int32_t pic_h);
/* The recorder's arrays from a caller-supplied allocator: the engine passes page-locked host memory so that the
 * per-picture flush (ovhip_job_flush) is plain asynchronous DMA out of the arrays the slots wrote. */
typedef struct ovhip_allocator {
    void *(*alloc)(void *user, size_t bytes);
    void  (*free)(void *user, void *p);
    void  *user;
} ovhip_allocator;
ovhip_recorder *ovhip_rec_create_ex(int32_t pic_w, int32_t pic_h, const ovhip_allocator *a);
/* on (default): ovhip_rec_dbf_ctu also maintains the dense edge planes of ovhip_rec_dbf_planes; off: edge lists only */
void  ovhip_rec_set_dense_dbf_planes(ovhip_recorder *rec, int on);
void  ovhip_rec_destroy(ovhip_recorder *rec);
void  ovhip_rec_reset(ovhip_recorder *rec);
/* Append the commands of one TU / PU.  Return number of commands appended or <0. */
int   ovhip_rec_tu(ovhip_recorder *rec, const ovhip_tu_state *st, const ovhip_tu_desc *tu);
/* A TU of a CU that has ordered tasks: intra_l / intra_c (either may be NULL) describe the luma / chroma prediction the
 * reference runs right before the TU's luma / chroma residual (rcn_intra_tu -> rcn_tu_st -> intra_pred_c,
 * rcn_transform_tree.c:1384-1430, :1269-1287; rcn_tu_c :1349-1382): fill x, y, log2_w, log2_h, kind, mode, the OVHIP_IF_CORNER /
 * MIP / BDPCM flags, avl_lft, avl_abv, mrl_idx, ciip_wt.  The recorder computes the level, stores the tasks in decoding
 * order, marks the TU's transform blocks OVHIP_RES_STORE and sets the tasks' OVHIP_IF_RES_* / scale fields.  A chroma
 * block that only needs an ordered chroma scale (inter CU below an ordered region) becomes an OVHIP_IT_RES_C task by itself.
 * Returns the number of transform-block commands appended or <0. */
int   ovhip_rec_tu_intra(ovhip_recorder *rec, const ovhip_tu_state *st, const ovhip_tu_desc *tu, const ovhip_itask *intra_l,
                         const ovhip_itask *intra_c);
/* tmp.recon_isp_subtree_v / _h (rcn_structures.h:480-491; rcn_transform_tree.c:1087-1205): an intra-sub-partition CU -- 2 or
 * 4 partitions side by side (vertical) or stacked; prediction and residual alternate partition by partition, so every
 * prediction call becomes an ordered task and every partition's transform block a STORE-mode command.  The partition
 * geometry, the transform types (DST-VII where mts_enabled and the side is 4..16) and the 1xN / 2xN / Nx1 / Nx2 block
 * paths are derived here as the reference derives them.  Vertical partitions narrower than 4 are predicted 4 columns at a
 * time (:1123-1138).  Returns the number of commands appended or <0. */
typedef struct ovhip_isp_desc {
    uint16_t x0, y0;                  /* luma position of the CU in the picture                                          */
    uint8_t  log2_cb_w, log2_cb_h;
    uint8_t  vertical;                /* 1: recon_isp_subtree_v, 0: recon_isp_subtree_h                                    */
    uint8_t  intra_mode;
    uint8_t  cbf_mask;                /* ISPTUInfo.cbf_mask: bit (nb_partitions - 1 - i) = partition i                     */
    uint8_t  lfnst_flag, lfnst_idx;
    uint8_t  mts_enabled;             /* ctudec->mts_enabled (:1110, :1180)                                                */
    /* per PREDICTION call, in call order (vertical: one per 4 columns; horizontal: one per partition): what fill_ref_left_0 /
     * fill_ref_above_0 read out of the progress bit-fields with the CU's geometry (rcn_intra.c:584-594) */
    uint8_t  corner[4];               /* bit 0: the above arm's map, bit 1: the left arm's map                                  */
    uint8_t  avl_abv[4], avl_lft[4];
    uint16_t last_pos[4];             /* ISPTUInfo.tb_info[i]                                                              */
    uint64_t sig_sb_map[4];
    const int16_t *coef;              /* ctudec->residual_y: partition i at i << (log2_tb_w + log2_tb_h)                   */
} ovhip_isp_desc;
int   ovhip_rec_isp_cu(ovhip_recorder *rec, const ovhip_tu_state *st, const ovhip_isp_desc *cu);
/* Number and size of the partitions and of the prediction calls of an ISP CU, as recon_isp_subtree_* derive them. */
void  ovhip_isp_geometry(int32_t log2_cb_w, int32_t log2_cb_h, int32_t vertical, int32_t *log2_pb, int32_t *n_pb, int32_t *log2_pred,
                         int32_t *n_pred);
/* The ordered tasks in decoding order, and sorted by level: level_start[l] .. level_start[l + 1] are the tasks of level
 * l + 1 (n_levels + 1 entries).  Sorting happens in the call. */
const ovhip_itask *ovhip_rec_itasks(const ovhip_recorder *rec, size_t *n);
const ovhip_itask *ovhip_rec_itasks_sorted(ovhip_recorder *rec, size_t *n, const uint32_t **level_start, uint32_t *n_levels);
/* The ordered tasks grouped by the CTU their block lies in (CTUs in raster order, inside a CTU by level, inside a level in
 * decoding order), with one descriptor per CTU that holds any.  deps: bit 0 left, 1 above-left, 2 above, 3 above-right
 * neighbour CTU must have finished its own tasks first: the union of its tasks' ctu_deps when every task carries them for
 * this CTU size, otherwise every such neighbour that holds ordered tasks at all. */
typedef struct ovhip_ictu {
    uint16_t cx, cy;          /* CTU column / row */
    uint32_t first, n;        /* its tasks in the returned array */
    uint32_t deps;
} ovhip_ictu;
/* CTU size (log2, 5..7; 7 when never called) the recorder relates ovhip_itask.ctu_deps to; call before the picture's first TU. */
int   ovhip_rec_set_ctu_size(ovhip_recorder *rec, int32_t log2_ctu_s);
uint32_t ovhip_rec_itask_levels(const ovhip_recorder *rec);      /* highest level recorded so far */
const ovhip_itask *ovhip_rec_itasks_by_ctu(ovhip_recorder *rec, int32_t log2_ctu_s, size_t *n, const ovhip_ictu **ctus, size_t *n_ctus);
/* tmp.rcn_transform_tree: walks the tree and records every leaf with ovhip_rec_tu.  Returns the number of
 * commands appended or <0. */
int   ovhip_rec_transform_tree(ovhip_recorder *rec, const ovhip_tu_state *st, const ovhip_tt_desc *tt);
int   ovhip_rec_pu(ovhip_recorder *rec, const ovhip_pu_desc *pu);
int   ovhip_rec_affine_cu(ovhip_recorder *rec, const ovhip_affine_desc *cu);
/* One inter coding unit in ONE call, whatever its kind (what a recorder-friendly caller hands over per CU, shim/caller.patch):
 * exactly one of pu (rcn_mcp_b; with refine flags the whole BDOF / DMVR coding unit, cut here as vcl_coding_unit.c:2450-2472 / :2598-2668
 * cut it; GPM) and aff (an affine coding unit) is non-NULL.  Returns what ovhip_rec_pu / ovhip_rec_affine_cu return. */
int   ovhip_rec_cu_inter(ovhip_recorder *rec, const ovhip_pu_desc *pu, const ovhip_affine_desc *aff);
/* rcn_ciip_weighted_sum: mode_abv / mode_lft = part_map.cu_mode_x[x_right >> log2_min_cb] /
 * cu_mode_y[y_bottom >> log2_min_cb] as enum CUMode (cu_utils.h:132-139). */
int   ovhip_rec_ciip(ovhip_recorder *rec, int32_t x0, int32_t y0, int32_t log2_w, int32_t log2_h,
                     int32_t mode_abv, int32_t mode_lft);
/* The CIIP weight 1 + (above CU intra) + (left CU intra) (rcn_inter.c:2977-2981) for ovhip_pu_desc.ciip_wt. */
int   ovhip_ciip_weight(int32_t mode_abv, int32_t mode_lft);
/* rcn_lmcs_compute_chroma_scale(lmcs_info, stride, progress_field, ctu_buff.y, x0, y0): abv_mask / lft_mask are
 * the two 16-bit availability masks it derives from progress_field (rcn_lmcs.c:327-332).  Returns the
 * region index; TUs recorded afterwards with lmcs_scale_c == 2 refer to it. */
int   ovhip_rec_lmcs_region(ovhip_recorder *rec, int32_t x0, int32_t y0, uint32_t abv_mask, uint32_t lft_mask);
/* Compact form of the edge planes: only the 4-sample segments that carry an edge, in raster order, luma
 * first, then Cb, then Cr (8 bytes each).  A lane of the device kernel then always has an edge to filter
 * (in the dense planes ~3/4 of the words are 0).  ux, uy: position in 4-luma-sample units. */
typedef struct ovhip_dbf_edge {
    uint16_t ux, uy;
    uint16_t word;            /* OVHIP_DBF_LUMA(...) or the chroma word, as in the planes          */
    uint8_t  comp;            /* 0 Y, 1 Cb, 2 Cr                                                   */
    uint8_t  pad;             /* offset-pair index, see ovhip_dbf_offsets; ovhip_dbf_compact writes 0 */
} ovhip_dbf_edge;
/* The deblocking offsets are slice-level state (DBFInfo.beta_offset / tc_offset = sh_luma_*_offset_div2 * 2,
 * slicedec.c:1416-1417): ovhip_rec_dbf_ctu gives every distinct pair of a picture an index, which travels in
 * ovhip_dbf_edge.pad.  More than OVHIP_DBF_MAX_OFFSETS distinct pairs in one picture: OVHIP_EUNSUP. */
#define OVHIP_DBF_MAX_OFFSETS 8
typedef struct ovhip_dbf_offsets { int8_t beta[OVHIP_DBF_MAX_OFFSETS], tc[OVHIP_DBF_MAX_OFFSETS]; } ovhip_dbf_offsets;
/* The edge lists ovhip_rec_dbf_ctu emitted so far (CTU by CTU, no sorting needed), dir 0 vertical / 1 horizontal. */
const ovhip_dbf_edge *ovhip_rec_dbf_edges(const ovhip_recorder *rec, int dir, size_t *n, ovhip_dbf_offsets *offsets);
/* dir 0: vertical edges, 1: horizontal.  planes: HOST pointers.  Writes at most cap entries to out (may be
 * NULL to count) and returns the number of edges, or <0. */
int64_t ovhip_dbf_compact(const ovhip_dbf_planes *planes, int dir, ovhip_dbf_edge *out, size_t cap);

/* Convert one CTU's deblocking maps into the picture-level edge planes.  Returns 0 or <0. */
int   ovhip_rec_dbf_ctu(ovhip_recorder *rec, const ovhip_dbf_ctu *ctu);
/* The same for n consecutive CTUs (a CTU row; n = 1 for a caller that keeps ONE struct DBFInfo, as the reference does) with the maps
 * read IN PLACE: ovhip_dbf_view = ovhip_dbf_ctu with every array by pointer -- into the caller's struct DBFInfo (ctudec.h:130-170;
 * same element layout: ctb_bound_* / aff_edg_* 49 words, bs* / affine_* 33 words, qp_* 34 x 33 bytes = struct DBFQPMap.hor) -- instead
 * of a 9 KB descriptor filled and copied per CTU.  ovhip_rec_dbf_mv_prepass_view: the MV-based pre-pass on the caller's
 * own bs1 maps (what the scalar slot does to dbf_info->bs1_map), no write-back copy. */
typedef struct ovhip_dbf_view {
    const uint64_t *ctb_bound_ver, *ctb_bound_hor, *ctb_bound_ver_c, *ctb_bound_hor_c, *aff_edg_ver, *aff_edg_hor;
    const uint64_t *bs2_ver, *bs2_hor, *bs2c_ver, *bs2c_hor, *bs1_ver, *bs1_hor, *bs1cb_ver, *bs1cb_hor, *bs1cr_ver, *bs1cr_hor, *affine_ver, *affine_hor;
    const uint8_t  *qp_y, *qp_cb, *qp_cr;
    int16_t  beta_offset, tc_offset;
    uint8_t  disable_v, disable_h, log2_ctu_s, last_x, last_y, ctu_lft, ctu_abv, pad;
    uint16_t ctu_w, ctu_h, ctb_x, ctb_y;
} ovhip_dbf_view;
int   ovhip_rec_dbf_row(ovhip_recorder *rec, const ovhip_dbf_view *ctus, size_t n);
int   ovhip_rec_dbf_mv_prepass_view(const ovhip_dbf_view *ctu, uint64_t *bs1_ver, uint64_t *bs1_hor, const ovhip_dbf_mv_ctx *mv);
/* Host copies of the edge planes (pointers valid until the next reset/destroy). */
int   ovhip_rec_dbf_planes(const ovhip_recorder *rec, ovhip_dbf_planes *out);
/* Bulk append of already-recorded commands to an EMPTY recorder (replay of a stored command stream). */
enum { OVHIP_REC_TB = 0, OVHIP_REC_COEF, OVHIP_REC_MC, OVHIP_REC_MCX, OVHIP_REC_AFF, OVHIP_REC_SIDE, OVHIP_REC_REGION,
       OVHIP_REC_CIIP, OVHIP_REC_EDGE_V, OVHIP_REC_EDGE_H, OVHIP_REC_ITASK };
int   ovhip_rec_append_raw(ovhip_recorder *rec, int which, const void *data, size_t n);
int   ovhip_rec_set_dbf_offsets(ovhip_recorder *rec, const ovhip_dbf_offsets *offsets, int n);
/* Access to the recorded (host) buffers. */
const ovhip_tb_cmd  *ovhip_rec_tb_cmds(const ovhip_recorder *rec, size_t *n);
/* The same commands reordered into four classes: big luma blocks, small luma blocks, big chroma blocks, small chroma
 * blocks (counts[0..3]); small = at most 256 samples and no side above 32 (16x16, 32x8, 8x32, 32x4, ...).  Luma first because with device-derived
 * chroma scales the chroma commands must run after ovhip_lmcs_scale_launch, which must run after the luma
 * ones; by size because ovhip_itx_launch_classes gives big and small blocks different workgroup shapes. */
const ovhip_tb_cmd  *ovhip_rec_tb_cmds_split(ovhip_recorder *rec, size_t counts[4], size_t *n);
const ovhip_lmcs_region *ovhip_rec_lmcs_regions(const ovhip_recorder *rec, size_t *n);
const int16_t       *ovhip_rec_coefs(const ovhip_recorder *rec, size_t *n_int16);
const ovhip_mc_unit *ovhip_rec_mc_units(const ovhip_recorder *rec, size_t *n);
/* The refined (OVHIP_MC_BDOF / OVHIP_MC_DMVR) units, kept apart so that each list is one launch. */
const ovhip_mc_unit *ovhip_rec_mcx_units(const ovhip_recorder *rec, size_t *n);
const ovhip_aff_unit *ovhip_rec_aff_units(const ovhip_recorder *rec, size_t *n);
const ovhip_ciip_unit *ovhip_rec_ciip_units(const ovhip_recorder *rec, size_t *n);
const int32_t        *ovhip_rec_aff_side(const ovhip_recorder *rec, size_t *n_int32);

/* ------------------------------------------------------------------------------------
 * Engine (device side).
 * ---------------------------------------------------------------------------------- */
typedef struct ovhip_ctx ovhip_ctx;

int  ovhip_abi_version(void);
/* stream: a hipStream_t the caller owns (e.g. torch's current stream) or NULL to create one. */
int  ovhip_ctx_create(ovhip_ctx **out, int device, void *stream);
void ovhip_ctx_destroy(ovhip_ctx *ctx);
/* 1 if the streams of the two (idle) contexts are served by the same hardware queue, 0 if not (measured: ~2 ms), < 0 error; and a
 * fresh stream for a context whose stream is in the wrong company (see ovvc_engine.hip) */
int  ovhip_ctx_shares_queue(ovhip_ctx *a, ovhip_ctx *b);
int  ovhip_ctx_new_stream(ovhip_ctx *ctx);
int  ovhip_ctx_sync(ovhip_ctx *ctx);
const char *ovhip_last_error(const ovhip_ctx *ctx);
/* Overlap of independent launches inside one stage: route the following launches to side stream k (1..3), or back
 * to the main stream (0); join makes the main stream wait for every side stream used since the last join. */
int  ovhip_ctx_fork(ovhip_ctx *ctx, int k);
int  ovhip_ctx_join(ovhip_ctx *ctx);
void *ovhip_ctx_stream(ovhip_ctx *ctx);

/* device memory helpers (plain hipMalloc/hipMemcpyAsync on the context stream) */
int  ovhip_malloc(ovhip_ctx *ctx, size_t bytes, void **dptr);
int  ovhip_free(ovhip_ctx *ctx, void *dptr);
int  ovhip_h2d(ovhip_ctx *ctx, void *dptr, const void *host, size_t bytes);
int  ovhip_d2h(ovhip_ctx *ctx, void *host, const void *dptr, size_t bytes);
int  ovhip_d2d(ovhip_ctx *ctx, void *dst, const void *src, size_t bytes);     /* synchronous */
/* page-locked host memory (DMA source / target: output frames, recorder arrays); NULL on failure */
void *ovhip_host_alloc(size_t bytes);
void  ovhip_host_free(void *p);
int  ovhip_pic_alloc(ovhip_ctx *ctx, int32_t w, int32_t h, ovhip_pic *pic);   /* three tight planes, zero-filled, complete on return */
int  ovhip_pic_free(ovhip_ctx *ctx, ovhip_pic *pic);
int  ovhip_pic_upload(ovhip_ctx *ctx, const ovhip_pic *pic, const uint16_t *y, const uint16_t *cb,
                      const uint16_t *cr, int32_t host_stride_y, int32_t host_stride_c);
int  ovhip_pic_download(ovhip_ctx *ctx, const ovhip_pic *pic, uint16_t *y, uint16_t *cb,
                        uint16_t *cr, int32_t host_stride_y, int32_t host_stride_c);

/* Stage launches.  cmds / coefs / units are DEVICE pointers; asynchronous on the ctx stream. */
int  ovhip_itx_launch(ovhip_ctx *ctx, const ovhip_pic *dst, const ovhip_tb_cmd *d_cmds,
                      uint32_t n_cmds, const int16_t *d_coefs, const int16_t *d_lmcs_scales);
/* Same, for a command list sorted by ovhip_rec_tb_cmds_split: the first n_large commands may have any size,
 * the following n_small commands must all be small in the sense of ovhip_rec_tb_cmds_split. */
int  ovhip_itx_launch_classes(ovhip_ctx *ctx, const ovhip_pic *dst, const ovhip_tb_cmd *d_cmds,
                              uint32_t n_large, uint32_t n_small, const int16_t *d_coefs,
                              const int16_t *d_lmcs_scales);
/* Same for a picture with ordered tasks: the OVHIP_RES_STORE commands write their residual (int16 bits) at the block's position
 * of `res`, a picture of dst's geometry (ovhip_pic_alloc(w, h)); the others add to dst as usual. */
int  ovhip_itx_launch_classes_res(ovhip_ctx *ctx, const ovhip_pic *dst, const ovhip_pic *res, const ovhip_tb_cmd *d_cmds,
                                  uint32_t n_large, uint32_t n_small, const int16_t *d_coefs, const int16_t *d_lmcs_scales);
/* The CHROMA commands of a picture (same classes) plus, riding in the same launch, the inverse LMCS mapping of the
 * luma plane (= ovhip_lmcs_inverse_launch).  Legal once the luma commands and ovhip_lmcs_scale_launch have run:
 * the commands must not address plane 0. */
int  ovhip_itx_launch_chroma_lmcs(ovhip_ctx *ctx, const ovhip_pic *dst, const ovhip_tb_cmd *d_cmds,
                                  uint32_t n_large, uint32_t n_small, const int16_t *d_coefs,
                                  const int16_t *d_lmcs_scales, const uint16_t *d_bwd_lut);
/* d_regions, d_scales: DEVICE; d_scales[i] receives lmcs_chroma_scale of region i.  luts: HOST. */
int  ovhip_lmcs_scale_launch(ovhip_ctx *ctx, const ovhip_pic *pic, const ovhip_lmcs_region *d_regions,
                             uint32_t n_regions, const ovhip_lmcs_luts *luts, int16_t *d_scales);
/* Maps the luma plane through d_bwd_lut (DEVICE, 1024 entries) in place. */
int  ovhip_lmcs_inverse_launch(ovhip_ctx *ctx, const ovhip_pic *pic, const uint16_t *d_bwd_lut);
/* intra: the picture holding the caller's planar prediction for units with a fused CIIP blend, or NULL. */
int  ovhip_mc_launch(ovhip_ctx *ctx, const ovhip_pic *dst, const ovhip_pic *refs, uint32_t n_refs,
                     const ovhip_mc_unit *d_units, uint32_t n_units, const uint16_t *d_lmcs_fwd_lut,
                     const ovhip_pic *intra);
/* Refined units (OVHIP_MC_BDOF / OVHIP_MC_DMVR).  d_mv_out: DEVICE array of 4 int32 per unit
 * (mv0x, mv0y, mv1x, mv1y finally used), or NULL.  It replaces the `OVMV *mv0, *mv1` in/out
 * arguments of rcn_dmvr_mv_refine (rcn_structures.h:628-632): the caller copies them into its
 * TMVP motion field as vcl_coding_unit.c:2621-2645 does. */
int  ovhip_mcx_launch(ovhip_ctx *ctx, const ovhip_pic *dst, const ovhip_pic *refs, uint32_t n_refs,
                      const ovhip_mc_unit *d_units, uint32_t n_units, const uint16_t *d_lmcs_fwd_lut,
                      int32_t *d_mv_out);
/* The decoder-side MV refinement of the units that carry OVHIP_MC_DMVR alone: refined vectors to d_mv_out (4 int32 per
 * unit of the list, other units' entries untouched), no sample written.  geom: any picture of the references' size. */
int  ovhip_dmvr_search_launch(ovhip_ctx *ctx, const ovhip_pic *geom, const ovhip_pic *refs, uint32_t n_refs,
                              const ovhip_mc_unit *d_units, uint32_t n_units, int32_t *d_mv_out);
/* CIIP: blends the intra prediction held in `intra` into `dst` (which holds the inter prediction). */
int  ovhip_ciip_launch(ovhip_ctx *ctx, const ovhip_pic *dst, const ovhip_pic *intra,
                       const ovhip_ciip_unit *d_units, uint32_t n_units);
/* Affine units; d_side: the DEVICE copy of ovhip_rec_aff_side(). */
int  ovhip_mca_launch(ovhip_ctx *ctx, const ovhip_pic *dst, const ovhip_pic *refs, uint32_t n_refs,
                      const ovhip_aff_unit *d_units, uint32_t n_units, const int32_t *d_side,
                      const uint16_t *d_lmcs_fwd_lut);
/* The refined and the affine units of a picture in ONE launch (they write disjoint blocks and read only the
 * references): equal to ovhip_mcx_launch + ovhip_mca_launch, one kernel boundary and one launch tail cheaper. */
int  ovhip_mcxa_launch(ovhip_ctx *ctx, const ovhip_pic *dst, const ovhip_pic *refs, uint32_t n_refs,
                       const ovhip_mc_unit *d_xunits, uint32_t n_xunits, int32_t *d_mv_out,
                       const ovhip_aff_unit *d_aunits, uint32_t n_aunits, const int32_t *d_side,
                       const uint16_t *d_lmcs_fwd_lut);
/* One LEVEL of the ordered pass (ovhip_itask): d_tasks = the n tasks of that level (DEVICE), all mutually independent; the
 * launch boundary to the next level makes their stores visible to it.  res: the residual picture the OVHIP_RES_STORE
 * commands wrote; d_regions / luts / d_scales as in ovhip_lmcs_scale_launch (NULL without LMCS chroma scaling): ordered regions
 * write their scale, scaled chroma residuals read it.  geom: the launch geometry (strips of the largest block, one or two
 * colour planes) from ovhip_intra_level_geom() on the HOST copy of the same tasks, or OVHIP_INTRA_GEOM_ANY (always valid,
 * launches up to 8x the workgroups, most of which leave at once). */
#define OVHIP_INTRA_GEOM_ANY 0x12u
uint32_t ovhip_intra_level_geom(const ovhip_itask *tasks, size_t n);
int  ovhip_intra_level_launch(ovhip_ctx *ctx, const ovhip_pic *pic, const ovhip_pic *res, const ovhip_itask *d_tasks, uint32_t n,
                              const ovhip_lmcs_region *d_regions, const ovhip_lmcs_luts *luts, int16_t *d_scales, int32_t log2_ctu_s,
                              uint32_t geom);
/* The WHOLE ordered pass in one launch: a workgroup per CTU that holds ordered tasks keeps the CTU's samples in LDS and runs
 * its tasks level by level; CTUs wait for their left / above-left / above / above-right neighbours (those that hold tasks)
 * through one flag word each, as the reference's wavefront threads do.  d_tasks / d_ctus: DEVICE copies of what
 * ovhip_rec_itasks_by_ctu() returned.  d_sync: ovhip_intra_sync_words() 32-bit words of device memory, zeroed once by its
 * owner; epoch: != 0 and different from every epoch this d_sync saw before (a picture counter).  The waits are bounded: on
 * expiry d_sync[0] becomes non-zero (1 + index of the CTU that gave up), the launch ends, the picture is incomplete -- the
 * owner must look at d_sync[0] after the launch; abort_mirror (may be NULL) is a second, device-writable address that receives the
 * same code, e.g. a page-locked host word (ovhip_job_wait reads that one and returns OVHIP_ELAUNCH).  Picture width must be a
 * multiple of 8, the planes 8-byte aligned. */
size_t ovhip_intra_sync_words(int32_t width, int32_t height, int32_t log2_ctu_s);
int  ovhip_intra_ctu_launch(ovhip_ctx *ctx, const ovhip_pic *pic, const ovhip_pic *res, const ovhip_itask *d_tasks, const ovhip_ictu *d_ctus,
                            uint32_t n_ctus, const ovhip_lmcs_region *d_regions, const ovhip_lmcs_luts *luts, int16_t *d_scales,
                            int32_t log2_ctu_s, uint32_t *d_sync, uint32_t epoch, uint32_t *abort_mirror);
/* The ordered pass in one launch, dependencies per 4x4 unit: every (task, strip, plane) item is a workgroup that polls the state
 * words of the units its reference arms cover, reads them with agent-scope loads, predicts, stores write-through and marks its
 * own units.  d_tasks: the LEVEL-SORTED tasks (ovhip_rec_itasks_sorted); d_items: ovhip_intra_flow_items() of that list (host
 * helper; 0 = the picture cannot take this path).  d_state: ovhip_intra_flow_words() words of device memory zeroed once;
 * epoch, abort_mirror and the bounded waits as for ovhip_intra_ctu_launch (d_state[0] = abort word).  The items may be launched
 * in several calls (consecutive ranges that end on level boundaries, same epoch): prepare != 0 only on the first, which marks the
 * units of ALL n_tasks tasks.  n_workers: the launch has that many workgroups and workgroup b takes the items b, b + n_workers, ...
 * in turn -- the bound on the pollers of this launch (wave slots the kernels of the other pictures in flight do not get, and what
 * has to fit the device beside the other flow launches for the forward-progress argument to hold); 0 or >= n_items: one workgroup
 * per item. */
size_t ovhip_intra_flow_words(int32_t width, int32_t height);
size_t ovhip_intra_flow_items(const ovhip_itask *sorted, size_t n, uint32_t *items, size_t cap);
int  ovhip_intra_flow_launch(ovhip_ctx *ctx, const ovhip_pic *pic, const ovhip_pic *res, const ovhip_itask *d_tasks, uint32_t n_tasks,
                             const uint32_t *d_items, uint32_t n_items, const ovhip_lmcs_region *d_regions, const ovhip_lmcs_luts *luts,
                             int16_t *d_scales, int32_t log2_ctu_s, uint32_t *d_state, uint32_t epoch, uint32_t *abort_mirror, int32_t prepare,
                             int32_t n_workers);
/* The flow launch hands samples from task to task with bit 15 set (kernels_intra.hip, FLOW_TAG).  This clears it in the blocks the
 * ordered tasks wrote: after the picture's flow launches, before anything else reads the picture.  with_luma == 0: chroma blocks
 * only -- ovhip_lmcs_inverse_launch drops the bit of every luma sample as a side effect of its table lookup.
 * On ENTRY to the flow launch no sample of `pic` may carry bit 15: true for every picture ovhip_pic_alloc returned (zero-filled)
 * that was only written by this library or by ovhip_pic_upload of real (<= 15-bit) samples since. */
int  ovhip_intra_flow_untag_launch(ovhip_ctx *ctx, const ovhip_pic *pic, const ovhip_itask *d_tasks, uint32_t n_tasks, int32_t with_luma);
/* ovhip_lmcs_scale_launch + the state words of the picture's flow launch (ovhip_intra_flow_launch's prepare step) in ONE launch:
 * d_tasks[n_tasks] = the level-sorted ordered tasks (DEVICE), d_state / epoch as for ovhip_intra_flow_launch, whose launches of
 * this picture are then called with prepare = 0. */
int  ovhip_lmcs_scale_prepare_launch(ovhip_ctx *ctx, const ovhip_pic *pic, const ovhip_lmcs_region *d_regions, uint32_t n_regions,
                                     const ovhip_lmcs_luts *luts, int16_t *d_scales, const ovhip_itask *d_tasks, uint32_t n_tasks,
                                     uint32_t *d_state, uint32_t epoch);
/* ovhip_lmcs_inverse_launch + ovhip_intra_flow_untag_launch(with_luma = 0) in ONE launch (a picture with LMCS whose ordered pass
 * ran as flow launches). */
int  ovhip_lmcs_inverse_untag_launch(ovhip_ctx *ctx, const ovhip_pic *pic, const uint16_t *d_bwd_lut, const ovhip_itask *d_tasks, uint32_t n_tasks);
/* planes->* are DEVICE pointers.  Filters `pic` in place: all vertical edges, then all horizontal. */
int  ovhip_dbf_launch(ovhip_ctx *ctx, const ovhip_pic *pic, const ovhip_dbf_planes *planes);
/* Same filter driven by the compact lists of ovhip_dbf_compact (DEVICE pointers). */
int  ovhip_dbf_launch_edges(ovhip_ctx *ctx, const ovhip_pic *pic, const ovhip_dbf_edge *d_edges_v, uint32_t n_v,
                            const ovhip_dbf_edge *d_edges_h, uint32_t n_h, int32_t beta_offset, int32_t tc_offset);
/* Same with the per-slice offset table of ovhip_rec_dbf_edges (edge.pad selects the pair). */
int  ovhip_dbf_launch_edges_ex(ovhip_ctx *ctx, const ovhip_pic *pic, const ovhip_dbf_edge *d_edges_v, uint32_t n_v,
                               const ovhip_dbf_edge *d_edges_h, uint32_t n_h, const ovhip_dbf_offsets *offsets);
/* d_params: DEVICE array of ceil(w/ctu)*ceil(h/ctu) entries.  dst and src must not alias. */
int  ovhip_sao_launch(ovhip_ctx *ctx, const ovhip_pic *dst, const ovhip_pic *src,
                      const ovhip_sao_ctu *d_params, int32_t log2_ctu_s);
/* Classification + luma / chroma ALF + CC-ALF.  dst and src must not alias. */
int  ovhip_alf_launch(ovhip_ctx *ctx, const ovhip_pic *dst, const ovhip_pic *src, const ovhip_alf_pic *alf);

/* ------------------------------------------------------------------------------------
 * Picture job: the per-picture "flush" in C.  One job = one picture in flight on one context (= one HIP stream, one
 * decoder frame thread): a recorder whose arrays are page-locked, the device copies of its buffers, and the launch
 * chain of the whole rcn path.  This is what the reference-side shim (shim/rcn_hip.c) calls from the LAST
 * alf.rcn_alf_filter_line of a picture (slicedec.c:940-955), before ovdpb_report_decoded_ctu_line publishes it:
 *
 *   ovhip_job_begin      rcn_attach_frame_buff (rcn_ctu.c:570): new picture, recorder reset
 *   ovhip_job_recorder   the recorder the slots append to while the picture is parsed
 *   ovhip_job_dmvr_rows  eager decoder-side MV refinement of the DMVR units recorded so far: search only, refined
 *                        vectors back on the host when it returns -- called from EVERY alf.rcn_alf_filter_line so that
 *                        the TMVP motion field of a CTU row is final before that row is published (the reference
 *                        stores the vectors right after each rcn_dmvr_mv_refine call, vcl_coding_unit.c:2621-2645)
 *   ovhip_job_flush      async H2D of commands / coefficients / edge lists / parameters, every stage launch, async D2H
 *                        of the refined vectors; returns without waiting
 *   ovhip_job_wait       blocks until the flush has completed on the device
 * ---------------------------------------------------------------------------------- */
typedef struct ovhip_job ovhip_job;

enum {                                   /* ovhip_job_params.stages (0 = all) */
    OVHIP_STAGE_MC = 1, OVHIP_STAGE_ITX = 2, OVHIP_STAGE_DBF = 4, OVHIP_STAGE_SAO = 8, OVHIP_STAGE_ALF = 16,
    OVHIP_STAGE_INTRA = 32,
    OVHIP_STAGE_INTRA_CTU = 0x20000000,  /* with OVHIP_STAGE_INTRA: the ordered pass as the one-launch CTU wavefront
                                          * (ovhip_intra_ctu_launch): measured slower than both others, DESIGN.md 4.1 */
    OVHIP_STAGE_INTRA_LEVELS = 0x10000000, /* with OVHIP_STAGE_INTRA: the ordered pass as one launch per level (ovhip_intra_level_launch)
                                            * instead of the default, one launch with per-unit dependency flags
                                            * (ovhip_intra_flow_launch; pictures it cannot take fall back to the levels) */
    OVHIP_STAGE_RESIDENT = 0x40000000    /* measurement only: no H2D / D2H, the device copies of the previous flush are replayed */
};

typedef struct ovhip_job_params {        /* picture-level side information; HOST pointers, copied by ovhip_job_flush */
    const ovhip_lmcs_luts *lmcs;         /* NULL: LMCS off                                                         */
    const ovhip_sao_ctu   *sao;          /* [n_ctu] raster, NULL: SAO off (stage skipped, dst stays deblocked)      */
    const ovhip_alf_ctu   *alf_ctus;     /* [n_ctu] raster, NULL: ALF off                                           */
    const int16_t *alf_luma_coeff, *alf_luma_clip;      /* [24][OVHIP_ALF_LUMA_SET_SIZE]                           */
    const int16_t *alf_chroma_coeff, *alf_chroma_clip;  /* [8][7]                                                  */
    const int16_t *alf_cc_coeff;                         /* [2][4][8]                                               */
    int32_t  log2_ctu_s;
    uint32_t stages;                     /* OVHIP_STAGE_* mask, 0 = all                                              */
    /* HIP events (hipEvent_t handles) the LAUNCH chain waits for on the job's stream, after the uploads have been enqueued:
     * the reference pictures (and the previous readers of dst) finishing on other streams.  The uploads do not depend on
     * them and so overlap the pictures this one waits for.  NULL / 0: none. */
    void *const *wait_events;
    uint32_t n_wait_events;
    /* Called by ovhip_job_flush on the flushing thread after the uploads have been enqueued and before the first launch: the
     * place to wait ON THE HOST for the reference pictures (what the reference's frame threads do, ovdpb_frame_synchro,
     * rcn_inter.c:131) while the uploads already run.  Unlike wait_events this leaves no barrier in the stream: a blocked
     * stream blocks the hardware queue it shares with other streams.  Non-zero return aborts the flush with OVHIP_EINVAL. */
    int (*before_launch)(void *user);
    void *before_launch_user;
    /* != 0: the flush also derives the TMVP plane cells of the refined units (ovhip_tmvp_cells_launch) and brings them back
     * with the refined vectors: ovhip_job_tmvp_cells().  nb_ctb_w of the plane = ceil(picture width / CTU size). */
    uint32_t tmvp_cells;
    /* != 0: wait_events are waited for ON THE HOST (hipEventSynchronize on the flushing thread, after the uploads have been
     * enqueued) instead of being put into the stream: before_launch without a callback into the caller's language. */
    uint32_t wait_on_host;
    /* Workers of a flow launch (ovhip_intra_flow_launch: n_workers); 0: the default -- 6 per compute unit, and never more than twice the
     * picture's widest level (an I picture: ~256).  The flow launches that run at once (one per hardware queue: 4 for a HIP process)
     * should fit the device, 16 per compute unit of that kernel; a launch that had to be abandoned halves the default for the
     * launches that follow.  A process that raises GPU_MAX_HW_QUEUES lowers this in proportion. */
    uint32_t flow_workers;
} ovhip_job_params;

typedef struct ovhip_job_stats {         /* what the last flush moved and launched */
    uint64_t h2d_bytes, d2h_bytes;
    uint32_t n_launches, n_h2d;
    uint32_t n_tb, n_mc, n_mcx, n_aff, n_edges_v, n_edges_h, n_regions, n_itasks, n_ilevels;
    uint32_t n_ordered_retries;          /* 1: ovhip_job_wait decoded the picture a second time, one launch per level (see there) */
    uint32_t flow_shift;                 /* the device's default worker count is 6 x CUs >> this: + 1 per abandoned flow launch (max 4), - 1 per 512 clean ones */
    /* host wall time of the flush call by phase, microseconds: class split + parameter block; enqueueing the copies; the
     * before_launch callback + wait events (the frame thread waiting for its reference pictures); enqueueing the launches */
    uint32_t host_us_prepare, host_us_upload, host_us_wait, host_us_launch;
} ovhip_job_stats;

int  ovhip_job_create(ovhip_ctx *ctx, int32_t pic_w, int32_t pic_h, ovhip_job **out);
void ovhip_job_destroy(ovhip_job *job);
ovhip_recorder *ovhip_job_recorder(ovhip_job *job);
/* Sizes every buffer of the job -- the recorder's page-locked arrays, the device copies, the staging blocks -- for a picture of the
 * job's size up front, so that no picture of a running decoder meets a growth (a growth of a device buffer is a device-wide
 * synchronisation, of a page-locked array an allocation + copy + free of 0.1-1.4 ms).  ovhip_frame_job() calls it for a frame thread's
 * job; ~30 MB page-locked + ~30 MB device memory at 4K. */
int  ovhip_job_reserve_for_picture(ovhip_job *job);
int  ovhip_rec_reserve_for_picture(ovhip_recorder *rec);
/* Waits until the previous flush no longer reads the recorder's arrays, then resets the recorder. */
int  ovhip_job_begin(ovhip_job *job);
/* Issues the job's next flushes on another context (= HIP stream) of the same device: a frame thread that became free takes
 * over a picture.  Waits for the job's previous flush first.  ctx == NULL: back to the context the job was created on -- a job must
 * not stay bound to a context that may be destroyed before it (ovhip_frame_submit returns a borrowed job this way). */
int  ovhip_job_bind(ovhip_job *job, ovhip_ctx *ctx);
/* dst: the picture being decoded; refs[n_refs]: the table ovhip_pu_desc.ref0/ref1 index; intra: picture with the
 * caller's planar prediction for fused CIIP blends, or NULL.  All DEVICE pictures of the job's size. */
int  ovhip_job_flush(ovhip_job *job, const ovhip_pic *dst, const ovhip_pic *refs, uint32_t n_refs,
                     const ovhip_pic *intra, const ovhip_job_params *params);
/* Waits for the flush.  If the ordered pass's flow launch gave up (bounded wait of a workgroup for its inputs: the launches of
 * several pictures can starve each other of compute-unit slots), the picture is decoded a second time right here with one
 * launch per level, from the recorder's arrays: ovhip_job_params' tables and the pictures passed to ovhip_job_flush must
 * stay valid until this call returns.  OVHIP_ELAUNCH only if that fails too. */
int  ovhip_job_wait(ovhip_job *job);
/* int32 [n][4] (mv0x, mv0y, mv1x, mv1y per refined unit, recorder order): valid after ovhip_job_wait, or, for the
 * units covered, after ovhip_job_dmvr_rows. */
const int32_t *ovhip_job_refined_mvs(ovhip_job *job, size_t *n_units);
/* Search-only pass over the refined units [first, current count) that carry OVHIP_MC_DMVR; synchronous: when it
 * returns the vectors are in ovhip_job_refined_mvs()[first..].  Returns the new `first` (= unit count) or <0. */
int64_t ovhip_job_dmvr_rows(ovhip_job *job, const ovhip_pic *refs, uint32_t n_refs);
/* The same pass in two halves, so that the search of one CTU row runs while the next row is parsed (slicedec.c:934-956 reports
 * row y - 1 after row y has been parsed: a pass begun at the end of row y - 1 is collected right before that report).
 *   _begin    enqueues: units [first, count) up, search, vectors down -- and, with log2_ctu_s != 0, the entries of the picture's
 *             collocated motion plane they belong to (ovhip_tmvp_cells_launch; 4 per unit, recorder order) -- then an event.
 *             Does not block.  Collects a pass still in flight first.  Returns the unit count it covers, or <0.
 *   _collect  waits for that event (one D2H of vectors + one of plane entries per row, both asynchronous until here); returns the
 *             number of units whose results are valid: ovhip_job_refined_mvs()[.. 4 n], ovhip_job_tmvp_cells()[.. 4 n].  0 passes
 *             pending: returns at once.  ovhip_job_flush collects by itself. */
int64_t ovhip_job_dmvr_rows_begin(ovhip_job *job, const ovhip_pic *refs, uint32_t n_refs, int32_t log2_ctu_s);
int64_t ovhip_job_dmvr_rows_begin_upto(ovhip_job *job, const ovhip_pic *refs, uint32_t n_refs, int32_t log2_ctu_s, size_t upto_units);   /* units [.., upto) only */
int64_t ovhip_job_dmvr_rows_collect(ovhip_job *job);
int  ovhip_job_last_stats(const ovhip_job *job, ovhip_job_stats *out);
/* ---- Band-wise submission: the picture enters the device while it is still being parsed (slicedec.c:815-975 reconstructs a CTU row
 * right after parsing it; dpb.c:1309-1323 reports it; a dependent picture runs a few rows behind, rcn_inter.c:131-146).
 *
 * ovhip_job_band() takes everything recorded since the previous call as one band of CTU rows ending at luma row `row_end` (a CTU-row
 * boundary; the recorder's arrays are in decoding order, so a band is a slice of every array) and enqueues, without waiting:
 *   the band's slices in ONE upload; its prediction, residuals and ordered pass; and the band's own filters: inverse luma mapping and
 *   deblocking of its rows (the horizontal edge on the boundary to the band ABOVE is in this band's lists: after the call the rows
 *   < row_end - 8 are final for the deblocking), then the SAO and ALF rows that made final -- all but the band's last 24 rows.  Intra
 *   prediction of the band BELOW reads this band's bottom row unfiltered (the reference's saved lines, rcn_ctu.c:246-510): the row is
 *   set aside before the filters and put back while the band below is reconstructed.  With `last` != 0 the call completes the picture
 *   (row_end = the picture's height).
 * upto: counts of the recorder's arrays that end the band (NULL: everything recorded so far -- the live decoder; a replay of a
 * recorded picture passes the counts at each CTU-row boundary).  refs: every picture the band's units read must be final in the rows
 * they reach -- the caller's business (ovhip_frame_band: the device DPB's row progress).  params: lmcs / log2_ctu_s / stages as for
 * ovhip_job_flush (the same in every call of a picture); sao / alf_* must be valid for the CTU rows the call's filters cover: rows
 * < row_end - 16 (SAO), < row_end - 24 (ALF), i.e. the CTU rows of the band itself -- parsed with the band (the shim takes them in
 * band_step: the reference's own ALF hook of a row runs a row later, slicedec.c:934-956).  Pictures with stand-alone CIIP units are
 * refused (OVHIP_EUNSUP): the shim never records them.
 * ovhip_job_wait() waits for everything enqueued; a picture whose ordered pass gave up FAILS (no second pass: its bands may have been
 * read).  ovhip_job_begin() starts the next picture as before.
 * ovhip_job_band_progress(): the picture rows [0, rows_final) are final once `event` (a hipEvent_t behind the last filter launch
 * enqueued so far; NULL: nothing yet) has completed and *abort_word (page-locked; the job's, for its lifetime) is still 0. */
typedef struct ovhip_band_counts { uint32_t n_tb, n_coef, n_mc, n_mcx, n_aff, n_side, n_reg, n_itask, n_edge_v, n_edge_h; } ovhip_band_counts;
void ovhip_rec_counts(const ovhip_recorder *rec, ovhip_band_counts *out);
int  ovhip_job_band(ovhip_job *job, const ovhip_pic *dst, const ovhip_pic *refs, uint32_t n_refs, const ovhip_job_params *params,
                    const ovhip_band_counts *upto, int32_t row_end, int32_t last);
int  ovhip_job_band_active(const ovhip_job *job);
int  ovhip_job_band_reserve(ovhip_job *job);        /* the band path's staging arena + side buffers now instead of in the first band */
int  ovhip_job_band_busy(ovhip_job *job);          /* 1: the last band's reconstruction is still running (never blocks) */
int  ovhip_job_band_progress(ovhip_job *job, int32_t *rows_final, void **event, const volatile uint32_t **abort_word);
/* row windows of the two frame-wide filters (rows: multiples of 64, or the picture's height) */
int  ovhip_sao_launch_rows(ovhip_ctx *ctx, const ovhip_pic *dst, const ovhip_pic *src, const ovhip_sao_ctu *d_params, int32_t log2_ctu_s,
                           int32_t row0, int32_t row1);
int  ovhip_alf_launch_rows(ovhip_ctx *ctx, const ovhip_pic *dst, const ovhip_pic *src, const ovhip_alf_pic *alf, int32_t row0, int32_t row1);

/* Measurement: bracket ONE launch group of every following flush with a HIP-event pair on the launch stream (stage -1:
 * off) and read back the accumulated duration.  A pair costs a few microseconds of stream time, hence one at a time. */
enum { OVHIP_TIME_MC = 0, OVHIP_TIME_MCXA, OVHIP_TIME_ITX_LUMA, OVHIP_TIME_LMCS_SCALE, OVHIP_TIME_ITX_CHROMA, OVHIP_TIME_DBF,
       OVHIP_TIME_SAO, OVHIP_TIME_ALF, OVHIP_TIME_INTRA, OVHIP_TIME_H2D, OVHIP_TIME_COUNT };
int  ovhip_job_time_stage(ovhip_job *job, int stage);
int  ovhip_job_stage_time(ovhip_job *job, double *sum_ms, uint64_t *count);

/* ------------------------------------------------------------------------------------
 * Output path.  Replaces the per-frame copy-out of examples/dectest.c:372-409
 * (write_decoded_frame_to_file): the conformance window is cropped and the three planes are packed, on the device,
 * into the byte layout that function writes -- 16-bit little-endian samples, the cropped Y rows, then Cb, then Cr, no
 * padding -- so that ONE contiguous D2H (or none, when only a digest is wanted) replaces three pitched plane copies.
 * `ovhip_window` = OVFrame.output_window (libovvc/ovframe.h): offsets in CHROMA sample units, as dectest.c:383-388
 * applies them (luma offsets are twice that).
 *
 * Digest: the reference's CI hashes the output FILE (CI/checkMD5.sh, md5sum); MD5 is a serial chain, so a whole frame
 * cannot be hashed by more than one lane, and a frame per lane of the host is ~40 ms at 4K.  Two things are offered instead:
 *   ovhip_pic_digest()           a per-picture FINGERPRINT, computed on the device, 16 bytes leaving it: a three-level hash tree over
 *                                the cropped frame -- leaf = mix128 of each 512-byte piece of a cropped row (the last piece of a row
 *                                shorter), row = mix128 of the row's leaf digests, band = mix128 of the digests of 8 consecutive
 *                                rows of one plane (the last band of a plane shorter), picture = MD5 of the band digests in the
 *                                order Y, Cb, Cr (that last step on the host).  mix128(words, tag): four 32-bit lanes seeded with
 *                                MD5's initial state (lane 0 xor tag), word i into lane i & 3 as h = (h ^ w) * 0x01000193 (FNV-1a),
 *                                murmur3's 32-bit finaliser per lane, then h0 += h1, h2 += h3, h0 += h2, h1 += h0, h2 += h0,
 *                                h3 += h0; words = two samples each, little endian; tag = samples / digests hashed.  (Until
 *                                round 4 every level was MD5: 6.5 M vector instructions per 4K picture.)  Any host can recompute
 *                                it from the written file (oracle/ovvc_oracle_output.py: picture_digest); it is NOT the md5sum of
 *                                the frame and not a cryptographic hash -- a change of any one sample changes it.
 *   ovhip_md5_* over ovhip_pic_output() frames    the plain host MD5 for callers that want the FILE's md5sum from the packed frames
 *                                (ovhip_stream_run with OVHIP_OUT_PACKED + OVHIP_STREAM_FILE_MD5 does exactly that).
 * ovhip_output_row_md5_launch (one plain MD5 per cropped row, one lane each) is the building block the first version of the
 * fingerprint used; it stays for callers that want per-row digests (~200 us per 4K picture: 120 chained blocks per lane).
 * ---------------------------------------------------------------------------------- */
typedef struct ovhip_window { uint16_t offset_lft, offset_rgt, offset_abv, offset_blw; } ovhip_window;
/* Bytes of the cropped frame / number of cropped rows (luma + 2 x chroma); 0 if the window leaves nothing. */
size_t ovhip_output_bytes(int32_t w, int32_t h, const ovhip_window *win);
size_t ovhip_output_rows(int32_t w, int32_t h, const ovhip_window *win);
/* d_out: DEVICE, ovhip_output_bytes() bytes.  Asynchronous on the ctx stream. */
int  ovhip_output_pack_launch(ovhip_ctx *ctx, const ovhip_pic *pic, const ovhip_window *win, uint16_t *d_out);
/* d_digests: DEVICE, 16 bytes per cropped row in the order Y rows, Cb rows, Cr rows.  Asynchronous. */
int  ovhip_output_row_md5_launch(ovhip_ctx *ctx, const ovhip_pic *pic, const ovhip_window *win, uint8_t *d_digests);
/* d_digests: DEVICE, 16 bytes per band (ovhip_output_bands()): the band level of the digest tree above.  Asynchronous. */
size_t ovhip_output_bands(int32_t w, int32_t h, const ovhip_window *win);
int  ovhip_output_tree_md5_launch(ovhip_ctx *ctx, const ovhip_pic *pic, const ovhip_window *win, uint8_t *d_digests);
/* Synchronous conveniences: pack + one D2H into host_dst (ovhip_output_bytes() bytes, pinned or pageable); the digest tree
 * + D2H of the band digests + MD5 over them into out[16]. */
int  ovhip_pic_output(ovhip_ctx *ctx, const ovhip_pic *pic, const ovhip_window *win, void *host_dst);
int  ovhip_pic_digest(ovhip_ctx *ctx, const ovhip_pic *pic, const ovhip_window *win, uint8_t out[16]);
/* Host MD5 (RFC 1321). */
typedef struct ovhip_md5_state { uint32_t h[4]; uint64_t n_bytes; uint8_t buf[64]; } ovhip_md5_state;
void ovhip_md5_init(ovhip_md5_state *st);
void ovhip_md5_update(ovhip_md5_state *st, const void *data, size_t n);
void ovhip_md5_final(ovhip_md5_state *st, uint8_t out[16]);

/* ------------------------------------------------------------------------------------
 * TMVP motion plane.  The reference's caller stores what rcn_dmvr_mv_refine returned into the CTU-local
 * 16x16 array of 8x8 cells tmvp_mv[l].mvs (vcl_coding_unit.c:2629-2645: cell ((x0 + 7) >> 3, (y0 + 7) >> 3) of the <= 16x16
 * block, its right neighbour for 16-wide and lower neighbour(s) for 16-high blocks), and tmvp_store_mv copies rows
 * 0 .. nb_tmvp_unit-1 of that array into the picture's plane (drv_lines.c:270-330: plane->mvs + ctb_offset + i * pln_stride,
 * pln_stride = nb_tmvp_unit * nb_ctb_w, nb_tmvp_unit = ctu >> 3).  ovhip_tmvp_cells_launch does that address arithmetic
 * on the device: for every refined unit with OVHIP_MC_DMVR it emits the plane cells the refined vectors belong in, in
 * PICTURE-LEVEL plane coordinates, so that the host applies a per-picture (or per-CTU-row) delta to the plane without
 * keeping per-CU bookkeeping -- the compressed-plane hand-over of the decoded picture's motion field.
 * out: 4 entries per unit (entry 4 * u + k; cell == OVHIP_TMVP_NONE: unused).
 * ---------------------------------------------------------------------------------- */
#define OVHIP_TMVP_NONE 0xffffffffu
typedef struct ovhip_tmvp_cell { uint32_t cell; int32_t mv0x, mv0y, mv1x, mv1y; } ovhip_tmvp_cell;
/* d_units / d_refined (4 int32 per unit: ovhip_mcx_launch's mv_out) / d_out: DEVICE.  Asynchronous on the ctx stream. */
int  ovhip_tmvp_cells_launch(ovhip_ctx *ctx, const ovhip_mc_unit *d_units, uint32_t n_units, const int32_t *d_refined,
                             int32_t log2_ctu_s, int32_t nb_ctb_w, ovhip_tmvp_cell *d_out);
/* After ovhip_job_wait of a flush with ovhip_job_params.tmvp_cells != 0: the cells of the picture's refined units (4 per
 * unit, recorder order), valid until the job's next begin. */
const ovhip_tmvp_cell *ovhip_job_tmvp_cells(ovhip_job *job, size_t *n_entries);


/* ====================================================================================
 * Frame threads and the device DPB (reference-independent; what shim/rcn_hip.c and the stream driver below are thin
 * callers of).
 *
 * The reference decodes pictures on frame threads (ovdec_select_subdec, ovdec.c:188-248); a picture's frame lives in the DPB
 * from ovdpb_init_picture until the last picture referencing it and the output process have dropped it (dpb.c), and a frame
 * thread that needs rows of a reference picture waits on that picture's progress (ovdpb_synchro_ref_decoded_ctus,
 * dpb.c:1242-1270; rcn_inter.c:131-146), which the producer reports per CTU row (ovdpb_report_decoded_ctu_line,
 * dpb.c:1309-1323; slicedec.c:934-956).  On the device a picture is decoded by ONE flush at the end of its parse, so the
 * progress mask collapses to one transition per picture:
 *
 *   ovhip_dpb_begin    (rcn_attach_frame_buff)        DECODING: a device picture for `key` on device `dev`
 *   ovhip_dpb_publish  (after ovhip_job_wait)         DONE / FAILED: every waiter is released.  ONLY ovhip_job_wait marks a
 *                                                     picture complete: it may decode the picture a second time (ordered pass),
 *                                                     so nothing recorded after ovhip_job_flush is a "picture done" signal
 *   ovhip_dpb_acquire  (ovdpb_frame_synchro)          blocks until `key` is DONE (OVHIP_EREF if it FAILED), pins it, returns
 *                                                     the picture as device `dev` sees it
 *   ovhip_dpb_unpin                                   the reader's own decode has completed
 *   ovhip_dpb_release  (frame unreferenced)           the slot and its device memory return to the pool once nobody has it pinned
 *
 * `key` is opaque (the shim passes the OVFrame pointer).  A key handed to ovhip_dpb_begin while an older picture still owns it
 * releases that picture first: the reference's frame pool re-uses an OVFrame only after every reference to it was dropped.
 *
 * Several devices in ONE process (north_star: frames shard one per GPU, mirroring --framethr): a picture is decoded on its
 * `home` device; a device whose queued pictures list it (ovhip_dpb_want, called when such a picture BEGINS, i.e. as soon as its
 * reference lists are known -- slicedec.c:1250-1256) receives a copy over xGMI as soon as it is DONE: hipMemcpyPeerAsync on the
 * destination device's copy stream, one event per copy; ovhip_dpb_acquire hands the event to the reader, which waits for it
 * after its own uploads are under way.  Devices that never list the picture never receive it.
 * ---------------------------------------------------------------------------------- */
typedef struct ovhip_dpb ovhip_dpb;

/* Memory / copy back-end of a DPB.  ovhip_dpb_create installs the HIP one; ovhip_dpb_create_ex lets a test supply its own
 * (the state machine is plain C + pthreads and is exercised without a GPU that way; it never computes samples). */
typedef struct ovhip_dpb_ops {
    void *user;
    int  (*pic_alloc)(void *user, int dev, int32_t w, int32_t h, ovhip_pic *pic);        /* zero-filled, complete on return    */
    void (*pic_free)(void *user, int dev, ovhip_pic *pic);
    /* src (home device src_dev, complete) -> dst (dst_dev).  *event: handle for copy_wait / copy_done, or NULL = done already */
    int  (*copy_start)(void *user, int dst_dev, const ovhip_pic *dst, int src_dev, const ovhip_pic *src, void **event);
    int  (*copy_wait)(void *user, int dst_dev, void *event);                               /* blocks the calling thread          */
    void (*copy_done)(void *user, int dst_dev, void *event);                               /* the handle is no longer needed     */
    /* a picture whose last decode FAILED goes back to the pool: make it safe to decode into (no sample may carry bit 15) */
    int  (*pic_clear)(void *user, int dev, const ovhip_pic *pic);
    /* row progress (ovhip_dpb_post_rows): the handles are the producer's (hipEvent_t behind a band's last filter launch); query: 1 =
     * completed, 0 = not yet, < 0 error; wait blocks the calling thread.  NULL (test back-ends): a posted record counts as completed. */
    int  (*event_query)(void *user, int dev, void *event);
    int  (*event_wait)(void *user, int dev, void *event);
} ovhip_dpb_ops;

#define OVHIP_MAX_DEVICES 16
typedef struct ovhip_dpb_stats {
    uint32_t n_live, n_pool;             /* pictures owned by a key (home + copies) / waiting in the free pools            */
    uint64_t n_begin, n_alloc, n_recycled, n_copies, copy_bytes, n_failed;
    uint64_t n_waits;                    /* acquire calls that had to block                                                */
} ovhip_dpb_stats;

/* devices[i] = HIP device ordinal of logical device i (the same ordinal may appear twice: two logical devices on one GPU,
 * the peer copy is then a device-to-device copy -- how the multi-device path is tested on a one-GPU box). */
int  ovhip_dpb_create(ovhip_dpb **out, const int *devices, int n_devices);
int  ovhip_dpb_create_ex(ovhip_dpb **out, int n_devices, const ovhip_dpb_ops *ops);
void ovhip_dpb_destroy(ovhip_dpb *d);            /* frees every device picture; no thread may be inside a DPB call */
int  ovhip_dpb_n_devices(const ovhip_dpb *d);
int  ovhip_dpb_device(const ovhip_dpb *d, int dev);                 /* HIP ordinal of logical device dev, <0 if none     */
int  ovhip_dpb_begin(ovhip_dpb *d, const void *key, int dev, int32_t w, int32_t h, ovhip_pic *pic);
int  ovhip_dpb_want(ovhip_dpb *d, const void *key, int dev);
/* The same with the caller's picture identity (`tag`, 0 = none; the shim: coded video sequence + POC of the OVPicture).  The frame
 * pool recycles OVFrame pointers and frame threads run on their own: a reader can ask for key F while F's slot still holds the
 * PREVIOUS picture that lived in F (DONE), because the thread decoding the new one has not reached rcn_attach_frame_buff yet.  With
 * tags, a slot whose tag differs from the reader's is "not begun yet": ovhip_dpb_acquire_tag waits for the right picture (bounded
 * like an unknown key), ovhip_dpb_want_tag is remembered until ovhip_dpb_begin_tag(key, tag) picks it up. */
int  ovhip_dpb_begin_tag(ovhip_dpb *d, const void *key, uint64_t tag, int dev, int32_t w, int32_t h, ovhip_pic *pic);
int  ovhip_dpb_want_tag(ovhip_dpb *d, const void *key, uint64_t tag, int dev);
int  ovhip_dpb_acquire_tag(ovhip_dpb *d, const void *key, uint64_t tag, int dev, ovhip_pic *pic, void **event);
/* status 0: DONE, else FAILED (latched error of the producer): every exit path of a producer publishes */
int  ovhip_dpb_publish(ovhip_dpb *d, const void *key, int status);
/* *event (may be NULL when dev is the home device): a copy_wait handle the caller waits for before it reads pic, or NULL.
 * A key nobody has begun YET is waited for as well (frame threads start in decoding order but run on their own), for at most
 * ovhip_dpb_set_unknown_key_timeout milliseconds (default 10000; 0: do not wait) -- then OVHIP_EINVAL. */
void ovhip_dpb_set_unknown_key_timeout(ovhip_dpb *d, int ms);
int  ovhip_dpb_acquire(ovhip_dpb *d, const void *key, int dev, ovhip_pic *pic, void **event);
/* 1: the picture is DONE (an acquire would not wait for its decode), 0: not yet (unknown key, another picture under it, DECODING),
 * OVHIP_EREF: it failed.  Never blocks. */
int  ovhip_dpb_poll_tag(ovhip_dpb *d, const void *key, uint64_t tag);
/* ---- row progress: the device analogue of ovdpb_report_decoded_ctu_line / ovdpb_synchro_ref_decoded_ctus (dpb.c:1309-1323, :1242-1270)
 * for the pictures a band-wise job decodes (ovhip_job_band).
 * ovhip_dpb_post_rows (producer, picture DECODING): the picture's rows [0, rows) are final once `event` has completed and
 *   *abort_word (NULL: none) still reads 0.  Records are kept in order; rows must not decrease.
 * ovhip_dpb_rows_tag (reader): are the rows [0, need_rows) of the picture there for device dev?  1: yes -- *pic is the picture on dev
 *   (pin != 0: pinned, as by ovhip_dpb_acquire_tag; unpin with ovhip_dpb_unpin), *event (may be NULL) a transfer the caller still has
 *   to wait for (ovhip_dpb_wait_copy) when the picture was decoded on another device; 0: not yet (never with block != 0); OVHIP_EREF:
 *   the picture FAILED or the DPB was shut down.  A picture decoded on ANOTHER device is there only when it is complete (its transfer
 *   is picture-granular); need_rows >= the picture's height means the whole picture.  block != 0 waits -- for the producer's records
 *   on the DPB's condition variable, for a record's event through event_wait. */
int  ovhip_dpb_post_rows(ovhip_dpb *d, const void *key, int32_t rows, void *event, const volatile uint32_t *abort_word);
int  ovhip_dpb_rows_tag(ovhip_dpb *d, const void *key, uint64_t tag, int dev, int32_t need_rows, int block, int pin, ovhip_pic *pic, void **event);
int  ovhip_dpb_wait_copy(ovhip_dpb *d, int dev, void *event);
int  ovhip_dpb_unpin(ovhip_dpb *d, const void *key);
int  ovhip_dpb_release(ovhip_dpb *d, const void *key);
/* The home picture of a DONE key, without blocking (output path): OVHIP_EINVAL unknown, OVHIP_EREF not (yet) decoded. */
int  ovhip_dpb_lookup(ovhip_dpb *d, const void *key, int *home_dev, ovhip_pic *pic);
/* Wakes every waiter with OVHIP_EREF and makes every later wait fail at once (decoder teardown after an error). */
void ovhip_dpb_shutdown(ovhip_dpb *d);
int  ovhip_dpb_get_stats(ovhip_dpb *d, ovhip_dpb_stats *out);

/* ------------------------------------------------------------------------------------
 * Frame thread: one context (HIP stream) + one picture job on one logical device of a DPB -- what the shim keeps per
 * OVCTUDec.  Call order per picture:
 *
 *   ovhip_frame_begin(key)          rcn_attach_frame_buff: DPB slot + ovhip_job_begin, empty reference table
 *   ovhip_frame_ref(ref_key)        first use of a reference picture: its index in the table the recorded units carry
 *                                   (ovhip_pu_desc.ref0 / ref1); tells the DPB this device wants it
 *   ... the slots record into ovhip_frame_recorder() ...
 *   ovhip_frame_dmvr_rows_collect() every alf.rcn_alf_filter_line: the refined vectors / plane entries of the rows parsed before
 *   ovhip_frame_dmvr_rows_begin()   the last one; then the pass over the row just parsed is enqueued (waits for the references
 *                                   first) and runs while the next row is parsed (ovhip_frame_dmvr_rows: both at once)
 *   ovhip_frame_submit(params)      last row: uploads; waits for the references (host) while they run; launches;
 *                                   ovhip_job_wait (incl. its second pass); THEN publishes -- on every exit path, with the
 *                                   error if there was one -- and unpins the references; then the optional output (the
 *                                   picture's readers do not wait for it)
 * ---------------------------------------------------------------------------------- */
typedef struct ovhip_frame ovhip_frame;

enum { OVHIP_OUT_NONE = 0,     /* the picture stays on the device (readers: ovhip_dpb_lookup + ovhip_pic_output / _digest) */
       OVHIP_OUT_DIGEST = 1,   /* + ovhip_pic_digest into ovhip_frame_output.digest (16 bytes leave the device)           */
       OVHIP_OUT_PLANES = 2,   /* + the three planes into caller memory (the OVFrame: dectest.c:372-409 reads it there)   */
       OVHIP_OUT_PACKED = 3 }; /* + crop + pack on the device, ONE D2H into caller memory (ovhip_output_bytes() bytes)    */
typedef struct ovhip_frame_output {
    int32_t  mode;                       /* OVHIP_OUT_*                                                                    */
    ovhip_window window;                 /* DIGEST / PACKED                                                                */
    uint16_t *y, *cb, *cr; int32_t stride_y, stride_c;    /* PLANES: host pointers, strides in samples                     */
    void    *packed;                     /* PACKED: host pointer                                                           */
    uint8_t  digest[16];                 /* DIGEST: result                                                                 */
} ovhip_frame_output;

/* Event trace of the frame layer: one record per ovhip_frame_* call that changes a picture's state, for every frame of the process
 * -- WHEN a caller (the shim's hooks under the decoder's events, slicedec.c:934-956) begins a picture, names its references, runs
 * the eager DMVR rows, submits.  a / b: BEGIN: logical device; REF: table index; DMVR_*: refined units recorded so far (b of
 * DMVR_BEGIN: the pass covers a DMVR unit = the references were waited for); SUBMIT: refined units recorded, reference count;
 * FAIL: status.  result = what the call returned.  On a DPB made with ovhip_dpb_create_ex (no device) ovhip_frame_create makes DRY
 * frames: the same state machine, DPB calls, waits and trace, nothing launched -- how the shim's device half runs in a container
 * without a GPU (oracle/ref_harness/gen_pipe.c "device" mode -> tests/golden/shim_pipe_dev.ovg, replayed on a GPU by
 * tests/test_gpu_pipe.py). */
enum { OVHIP_FE_BEGIN = 1, OVHIP_FE_REF, OVHIP_FE_DMVR_ROWS, OVHIP_FE_DMVR_BEGIN, OVHIP_FE_DMVR_COLLECT, OVHIP_FE_SUBMIT, OVHIP_FE_FAIL,
       OVHIP_FE_BAND /* a = row_end, b = last, result: 1 the band went to the device, 0 left to the next call, < 0 error */ };
typedef struct ovhip_frame_event {
    uint32_t op; int32_t frame;          /* OVHIP_FE_*; the frame object, numbered in creation order                      */
    uint64_t key, tag;
    int64_t  a, b, result;
} ovhip_frame_event;
void ovhip_frame_set_trace(void (*sink)(void *user, const ovhip_frame_event *ev), void *user);    /* NULL: off */

int  ovhip_frame_create(ovhip_dpb *dpb, int dev, int32_t w, int32_t h, ovhip_frame **out);
void ovhip_frame_destroy(ovhip_frame *f);
ovhip_ctx      *ovhip_frame_ctx(ovhip_frame *f);
ovhip_job      *ovhip_frame_job(ovhip_frame *f);
ovhip_recorder *ovhip_frame_recorder(ovhip_frame *f);
int  ovhip_frame_begin(ovhip_frame *f, const void *key);
int  ovhip_frame_ref(ovhip_frame *f, const void *ref_key);          /* index (0..15) or <0 */
int  ovhip_frame_begin_tag(ovhip_frame *f, const void *key, uint64_t tag);       /* with the picture identity: ovhip_dpb_begin_tag */
int  ovhip_frame_ref_tag(ovhip_frame *f, const void *ref_key, uint64_t tag);
/* Same without the search for an existing entry: the table entry `slot` (= the number of entries so far) is ref_key, which may
 * already sit in another entry (a recorded picture whose units index a fixed table). */
int  ovhip_frame_ref_at(ovhip_frame *f, int slot, const void *ref_key);
int64_t ovhip_frame_dmvr_rows(ovhip_frame *f);
/* The two halves (ovhip_job_dmvr_rows_begin / _collect): _begin waits for the picture's reference pictures on the host (as
 * ovhip_frame_dmvr_rows does: only a published picture is complete, its ordered pass may run a second time), then enqueues the pass
 * over the units recorded so far and returns; _collect before the row those units belong to is reported decoded. */
/* 1: every reference picture named so far is complete (and now pinned: _begin / _submit will not wait for a decode), 0: not yet, < 0: one
 * failed.  Never waits for a decode: a caller that can go on parsing asks this before it starts a row pass. */
int  ovhip_frame_refs_ready(ovhip_frame *f);
int64_t ovhip_frame_dmvr_rows_begin(ovhip_frame *f, int32_t log2_ctu_s);
int64_t ovhip_frame_dmvr_rows_begin_upto(ovhip_frame *f, int32_t log2_ctu_s, size_t upto_units);
int64_t ovhip_frame_dmvr_rows_collect(ovhip_frame *f);
/* job: NULL = the frame's own job (the shim); else a job holding an already recorded picture of the same size, which is bound to
 * this frame's context for the flush (the stream driver's pre-recorded pictures).  intra: as ovhip_job_flush.  out: NULL =
 * OVHIP_OUT_NONE.  Returns the picture's status (what was published). */
int  ovhip_frame_submit(ovhip_frame *f, ovhip_job *job, const ovhip_pic *intra, const ovhip_job_params *params, ovhip_frame_output *out);
/* A picture that cannot be submitted (latched recorder error, unsupported tool): publishes it as FAILED so that no reader
 * waits for it for ever. */
/* ---- band-wise submission (the picture enters the device while it is parsed; see ovhip_job_band) ----
 * ovhip_frame_set_band_mode(f, 1): ovhip_frame_refs_ready / ovhip_frame_dmvr_rows_begin then ask the device DPB for the ROWS the units
 *   in question read (ovhip_dpb_rows_tag) instead of whole reference pictures.
 * ovhip_frame_band(f, params, row_end, last, out): everything recorded since the last band that went, as a band ending at the CTU-row
 *   boundary row_end.  Returns 1: enqueued (and the rows it made final posted to the DPB), 0: a reference picture does not have the rows
 *   yet -- nothing was enqueued, the next call takes this band's units too (never with last != 0, which waits), < 0: error.  With last
 *   != 0 the call completes the picture exactly as ovhip_frame_submit does: wait, publish, output. */
int  ovhip_frame_set_band_mode(ovhip_frame *f, int on);
int  ovhip_frame_band(ovhip_frame *f, const ovhip_job_params *params, int32_t row_end, int32_t last, ovhip_frame_output *out);
/* The same for a picture whose parse is AHEAD of its reference pictures (it has been recorded further than row_end): the band ends at the
 * recorder's array lengths `upto` (ovhip_rec_counts taken when row_end had just been parsed), and with block != 0 the call waits for the
 * reference rows instead of returning 0 -- how the picture's last hook works through the rows its references had not reached while it was
 * parsed, row by row as they arrive, so that ITS rows reach ITS readers as early (never the picture's last band: use ovhip_frame_band). */
int  ovhip_frame_band_upto(ovhip_frame *f, const ovhip_job_params *params, int32_t row_end, const ovhip_band_counts *upto, int32_t block);
int  ovhip_frame_band_stats(const ovhip_frame *f, int32_t *n_bands, int32_t *n_deferred);
int  ovhip_frame_fail(ovhip_frame *f, int status);
const char *ovhip_frame_last_error(const ovhip_frame *f);

/* ------------------------------------------------------------------------------------
 * Call log: the arguments of the recorder entry points of one picture (descriptors + the coefficient blocks they point to,
 * in the reference's layout), serialised in call order -- the compact pre-parsed form of a picture.  ovhip_calllog_replay
 * issues the same calls again: what a parse thread does per picture minus CABAC, used by the stream driver to put the
 * recorder into the timed region and by tools/micro/rec_throughput.c.
 * ---------------------------------------------------------------------------------- */
typedef struct ovhip_calllog ovhip_calllog;
ovhip_calllog *ovhip_calllog_create(void);
void   ovhip_calllog_destroy(ovhip_calllog *log);
void   ovhip_calllog_reset(ovhip_calllog *log);
const void *ovhip_calllog_data(const ovhip_calllog *log, size_t *bytes);
/* While a log is attached every ovhip_rec_tu / _tu_intra / _isp_cu / _pu / _affine_cu / _lmcs_region / _dbf_ctu /
 * _set_ctu_size call on `rec` is appended to it (NULL detaches). */
void   ovhip_rec_set_calllog(ovhip_recorder *rec, ovhip_calllog *log);
/* Re-issues the calls of a serialised log on rec.  Returns the number of calls or <0 (first failing call's code). */
int64_t ovhip_calllog_replay(const void *data, size_t bytes, ovhip_recorder *rec);

/* ------------------------------------------------------------------------------------
 * Stream driver: decodes a whole stream of recorded pictures with N frame threads per device, in C (pthreads) -- the frame
 * thread pool of the reference (ovdec.c:188-248, --framethr) for the device path, and the timed region of bench.py.
 * Pictures are listed in DECODING order; a free thread of a picture's device takes the next picture of that device, records
 * it (OVHIP_STREAM_RECORD: replay of its call log into the thread's own job, as a parse thread would) or takes its
 * pre-recorded job, waits for its reference pictures (DPB), submits, publishes.  A picture is released from the DPB when the
 * last picture that lists it and the output thread are done with it.
 * ---------------------------------------------------------------------------------- */
#define OVHIP_STREAM_MAX_REFS 8
typedef struct ovhip_stream_content {      /* one recorded picture, shared by every stream picture that shows it */
    const void *calllog; size_t calllog_bytes;        /* OVHIP_STREAM_RECORD                                         */
    ovhip_job_params params;                          /* picture-level tables (host pointers, stay valid)            */
    uint32_t n_ref_slots;                             /* size of the reference table its units index (ref0 / ref1)    */
} ovhip_stream_content;

typedef struct ovhip_stream_pic {
    uint32_t content;                      /* index into contents[]                                                 */
    uint32_t job;                          /* pre-recorded mode: index into jobs[] (a job is in flight once at a time) */
    int32_t  poc;                          /* output order                                                          */
    uint16_t device;                       /* logical device it is decoded on                                       */
    uint16_t n_refs;
    uint32_t refs[OVHIP_STREAM_MAX_REFS];  /* indices (decoding order, < own) of its reference pictures: table slot k = refs[k % n_refs] */
    int32_t  owner;                        /* rank that decodes it (multi-process); != rank: arrives through xfer.recv   */
    uint32_t send_mask;                    /* ranks (bit r) it is sent to after decoding                            */
} ovhip_stream_pic;

typedef struct ovhip_stream_xfer {         /* multi-process exchange (one process per GPU): callbacks of the host harness */
    void *user;
    int (*send)(void *user, uint32_t idx, const ovhip_pic *pic, int dst_rank);     /* returns when the buffer may be reused */
    int (*recv)(void *user, uint32_t idx, const ovhip_pic *pic, int src_rank);     /* returns when the data is in place     */
} ovhip_stream_xfer;

/* RCCL transport (ovvc_rccl.hip): a ready-made ovhip_stream_xfer for one process per GPU -- the three planes of a picture as ncclSend /
 * ncclRecv in ONE ncclGroup per picture on a stream of its own, issued by the stream driver's communication thread; no collective.
 * librccl.so is opened at run time.  The launcher hands every rank the 128-byte ncclUniqueId rank 0 made (ovhip_rccl_unique_id),
 * e.g. through torch.distributed's broadcast (bench.py) or MPI.  ovhip_rccl_self_exchange: a one-rank check of the same path. */
typedef struct ovhip_rccl ovhip_rccl;
int  ovhip_rccl_unique_id(uint8_t out[128]);
int  ovhip_rccl_create(ovhip_rccl **out, const uint8_t unique_id[128], int rank, int world, int hip_device);
void ovhip_rccl_destroy(ovhip_rccl *r);
const ovhip_stream_xfer *ovhip_rccl_xfer(ovhip_rccl *r);
const char *ovhip_rccl_last_error(const ovhip_rccl *r);
int  ovhip_rccl_stats(const ovhip_rccl *r, uint64_t out[4]);      /* pictures sent, bytes sent, pictures received, bytes received */
int  ovhip_rccl_self_exchange(ovhip_rccl *r, const ovhip_pic *src, const ovhip_pic *dst);

enum { OVHIP_STREAM_RECORD = 1,            /* record every picture from its call log inside the run (else: pre-recorded jobs) */
       OVHIP_STREAM_DIGESTS = 2,           /* per-picture ovhip_pic_digest into result digests (16 bytes per picture)        */
       OVHIP_STREAM_RESIDENT = 4,          /* measurement: flushes replay the device copies (OVHIP_STAGE_RESIDENT)           */
       OVHIP_STREAM_KEEP = 8,              /* do not release pictures at the end of the run (the next run continues the stream) */
       OVHIP_STREAM_HOLD_ALL = 32,         /* (with OVHIP_STREAM_KEEP, given to the run that starts the stream) no picture is released before the
                                            * stream ends: every picture stays readable through ovhip_stream_key (tests)           */
       OVHIP_STREAM_FILE_MD5 = 16 };       /* OVHIP_OUT_PACKED: also hash the frames on the host (MD5 runs at ~0.6 GB/s per core: 40 ms per 4K
                                            * frame -- for conformance checks, not for throughput runs)                          */

typedef struct ovhip_stream_cfg {
    int32_t w, h;
    uint32_t flags;                        /* OVHIP_STREAM_*                                                         */
    int32_t threads_per_device;            /* frame threads (= pictures in flight) per logical device               */
    int32_t output;                        /* OVHIP_OUT_NONE / _DIGEST / _PACKED: an output thread takes the pictures in POC order */
    ovhip_window window;
    uint32_t extra_stages;                 /* OR-ed into every flush's stage mask (OVHIP_STAGE_INTRA_LEVELS ...)     */
    int32_t rank;                          /* this process's rank (pictures with owner != rank are received)         */
    const ovhip_stream_xfer *xfer;         /* NULL: single process                                                   */
    /* > 0: one more frame thread per device that takes pictures WITHOUT reference pictures (intra pictures) up to this many
     * pictures before their turn in decoding order -- such a picture is a dependency chain of milliseconds (ordered pass) that
     * everything after it waits for; started early it runs beside the pictures before it.  One extra picture buffer, no latency. */
    int32_t intra_lookahead;
    int32_t ahead_own_queue;               /* != 0: at creation, streams of in-order threads that share that thread's hardware queue are replaced
                                            * until none does (ovhip_ctx_shares_queue; ~2 ms per probe)                                     */
} ovhip_stream_cfg;

typedef struct ovhip_stream_result {
    double   seconds;                      /* first picture taken .. last picture published and devices idle        */
    uint64_t n_decoded, n_second_passes, n_received, n_sent;
    uint64_t out_frames, out_bytes;
    uint8_t  out_md5[16];                  /* OVHIP_OUT_PACKED + OVHIP_STREAM_FILE_MD5: MD5 of the concatenated frames in output order
                                            * = md5sum of the file dectest would write (CI/checkMD5.sh); OVHIP_OUT_DIGEST: MD5
                                            * over the pictures' digests (a private fingerprint)                            */
    double   record_seconds;               /* OVHIP_STREAM_RECORD: time spent replaying call logs, summed over threads */
    /* host seconds summed over the frame threads: the four phases of ovhip_job_stats.host_us_*, then inside ovhip_job_wait */
    double   host_seconds[5];
    int32_t  status;                       /* 0 or the first error                                                   */
    char     error[192];
    /* in: NULL, or room for 8 doubles per picture of the run -- seconds since the run began at which the picture was taken by a
     * frame thread, entered ovhip_frame_submit, was published (left it); the thread's index; then: its reference pictures were in
     * the thread's hands (uploads enqueued, references acquired), its launches were enqueued, ovhip_frame_submit returned (after
     * the output); one spare (analysis of stalls: tools/debug/dep_latency.py) */
    double  *trace;
} ovhip_stream_result;

typedef struct ovhip_stream ovhip_stream;
/* jobs: pre-recorded pictures (n_jobs may be 0 with OVHIP_STREAM_RECORD in every run).  The DPB, contents, jobs stay the
 * caller's.  Creates threads_per_device frames per device. */
int  ovhip_stream_create(ovhip_stream **out, ovhip_dpb *dpb, const ovhip_stream_cfg *cfg, const ovhip_stream_content *contents,
                         uint32_t n_contents, ovhip_job *const *jobs, uint32_t n_jobs);
void ovhip_stream_destroy(ovhip_stream *s);
/* Decodes pics[first .. first + n) of the stream pics[0 .. n_total) (indices and refs are positions in `pics`).  A call with
 * the same pics / n_total and first != 0 CONTINUES the stream of the previous call (which must have run with OVHIP_STREAM_KEEP:
 * its pictures are still in the DPB); anything else starts a new stream.  digests: n * 16 bytes or NULL.  flags: OVHIP_STREAM_*
 * of this run, OR-ed with the configuration's. */
int  ovhip_stream_run(ovhip_stream *s, const ovhip_stream_pic *pics, uint32_t n_total, uint32_t first, uint32_t n, uint32_t flags,
                      uint8_t *digests, ovhip_stream_result *res);
ovhip_frame *ovhip_stream_frame(ovhip_stream *s, int dev, int thread);
/* ahead_own_queue: streams replaced at creation / in-order streams that still share the look-ahead thread's hardware queue */
int  ovhip_stream_queue_info(const ovhip_stream *s, int *moved, int *sharing);
/* The DPB key of picture idx of the current stream (valid while the run that decoded it kept it: OVHIP_STREAM_KEEP). */
const void *ovhip_stream_key(const ovhip_stream *s, uint32_t idx);
/* test hook: the next flush of `job` that has a flow launch is ABORTED for real (the device's abort word is set before the
 * launch, so the first pass leaves the picture incomplete) -- ovhip_job_wait must then produce the picture with its second pass */
int  ovhip_job_test_abort_next_flow(ovhip_job *job);

#ifdef __cplusplus
}
#endif
#endif /* OVVC_HIP_H */
