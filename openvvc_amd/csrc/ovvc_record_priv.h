/* ovvc_record_priv.h -- recorder state shared by ovvc_record.c / ovvc_record_dbf.c (private). */
#ifndef OVVC_RECORD_PRIV_H
#define OVVC_RECORD_PRIV_H
#include <stddef.h>
#include "ovvc_hip.h"

struct ovhip_recorder {
    int32_t pic_w, pic_h;
    ovhip_allocator al; int has_al;     /* arrays below come from it (pinned host memory in the engine) */
    ovhip_tb_cmd  *tb;    size_t n_tb,   cap_tb;
    int16_t       *coef;  size_t n_coef, cap_coef;
    ovhip_mc_unit *mc;    size_t n_mc,   cap_mc;
    ovhip_mc_unit *mcx;   size_t n_mcx,  cap_mcx;   /* BDOF / DMVR units */
    ovhip_aff_unit *aff;  size_t n_aff,  cap_aff;   /* affine units */
    int32_t *aff_side;    size_t n_side, cap_side;
    ovhip_lmcs_region *reg; size_t n_reg, cap_reg;
    ovhip_ciip_unit *ciip; size_t n_ciip, cap_ciip;
    ovhip_tb_cmd *tb_split; size_t cap_split;
    /* deblocking edge planes (ovvc_record_dbf.c) */
    uint16_t *dbf_luma_v, *dbf_luma_h, *dbf_cb_v, *dbf_cr_v, *dbf_cb_h, *dbf_cr_h;
    int32_t dbf_w4, dbf_h4;
    int16_t dbf_beta_offset, dbf_tc_offset;
    int dense_planes;                   /* also maintain the dense planes above (ovhip_rec_dbf_planes) */
    /* compact edge lists, emitted directly per CTU (what the device kernel consumes) */
    ovhip_dbf_edge *edge_v, *edge_h; size_t n_edge_v, cap_edge_v, n_edge_h, cap_edge_h;
    ovhip_dbf_offsets dbf_off; int n_dbf_off;   /* distinct (beta, tc) offset pairs of the picture's slices */
    /* ordered tasks (ovvc_record_intra.c) */
    ovhip_itask *itask; size_t n_itask, cap_itask;
    ovhip_itask *itask_sorted; size_t cap_isorted;
    uint32_t *ilevel_start; size_t cap_ilevel; uint32_t n_ilevels, max_ilevel;
    ovhip_itask *itask_ctu; size_t cap_itask_ctu;       /* grouped by CTU (ovhip_rec_itasks_by_ctu) */
    ovhip_ictu *ictu; size_t cap_ictu;
    uint32_t *ctu_count; size_t cap_ctu_count;
    uint16_t *lvl_y, *lvl_c;            /* level of the ordered task covering each 4x4-luma unit (0: none), luma / chroma */
    int32_t lvl_w4, lvl_h4; int lvl_dirty;
    int log2_ctu;                       /* CTU size the tasks' ctu_deps refer to (ovhip_rec_set_ctu_size; 7 by default) */
    int scan_cx, scan_cy; uint32_t scan_deps, region_deps;
    uint16_t *reg_level; size_t cap_reglvl;     /* level of each chroma-scale region (0: derived by the unordered launch) */
    ovhip_calllog *log;                 /* ovhip_rec_set_calllog: every entry-point call is also serialised there */
};

/* ovvc_calllog.c */
void ovhip_calllog_tu_(ovhip_calllog *l, const ovhip_tu_state *st, const ovhip_tu_desc *tu, const ovhip_itask *il, const ovhip_itask *ic);
void ovhip_calllog_isp_(ovhip_calllog *l, const ovhip_tu_state *st, const ovhip_isp_desc *cu);
void ovhip_calllog_pu_(ovhip_calllog *l, const ovhip_pu_desc *pu);
void ovhip_calllog_affine_(ovhip_calllog *l, const ovhip_affine_desc *cu);
void ovhip_calllog_region_(ovhip_calllog *l, int32_t x0, int32_t y0, uint32_t abv, uint32_t lft);
void ovhip_calllog_dbf_(ovhip_calllog *l, const ovhip_dbf_ctu *c);
void ovhip_calllog_ciip_(ovhip_calllog *l, int32_t x0, int32_t y0, int32_t log2_w, int32_t log2_h, int32_t mode_abv, int32_t mode_lft);
void ovhip_calllog_ctu_size_(ovhip_calllog *l, int32_t log2_ctu_s);

int  ovhip_rec_intra_reset_(ovhip_recorder *r);
void ovhip_rec_intra_free_(ovhip_recorder *r);
/* recorder-internal: the ordered tasks of one TU, called by ovhip_rec_tu_intra around its transform blocks */
int  ovhip_rec_itask_add_(ovhip_recorder *r, const ovhip_itask *t, uint16_t extra_level);
uint16_t ovhip_rec_region_level_(ovhip_recorder *r, int32_t x0, int32_t y0, int n_abv, int n_lft);

int  ovhip_rec_grow_(ovhip_recorder *r, void **p, size_t *cap, size_t need, size_t elem);
void ovhip_rec_free_(ovhip_recorder *r, void *p);

/* band-wise submission (ovvc_picture.hip): class split / level sort of a RANGE of the recorded commands into the caller's buffer */
void ovhip_rec_tb_split_range_(const ovhip_recorder *r, size_t first, size_t n, ovhip_tb_cmd *out, size_t counts[4], size_t tiny[4][4]);
int  ovhip_rec_itasks_sorted_range_(const ovhip_recorder *r, size_t first, size_t n, ovhip_itask *out, uint32_t *level_start, size_t cap, uint32_t *n_levels);

void ovhip_rec_dbf_reset_(ovhip_recorder *r);
void ovhip_rec_dbf_free_(ovhip_recorder *r);
#endif
