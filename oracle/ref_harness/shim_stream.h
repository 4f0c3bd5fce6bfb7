/* shim_stream.h -- TEST INFRASTRUCTURE (this container only): second mode of gen_golden ("shim").
 *
 * In this mode the harness installs the MI355X override block (shim/rcn_hip.c, rcn_init_functions_hip) on top of the
 * scalar table and drives the INSTALLED slots with exactly the seeded OVCTUDec states of the reference run (same seeds,
 * same draw order).  The slots record; what they recorded is written to tests/golden/shim_*.ovg as command streams,
 * one slice per case.  tests/ then execute those streams (oracle on CPU, HIP kernels on the GPU) and compare with the
 * bytes the REFERENCE produced for the same case in tests/golden/*.ovg: this pins the ctudec -> descriptor mapping of
 * the shim itself, not a restatement of it in test code.
 */
#ifndef SHIM_STREAM_H
#define SHIM_STREAM_H
#include "rcn_hip.h"

static int g_shim;

#define SHIM_NARR 11            /* OVHIP_REC_TB .. OVHIP_REC_ITASK */
struct shim_stream { gbuf arr[SHIM_NARR]; gbuf off; uint32_t n_cases; gbuf mv_chk; gbuf tmvp_exp; };

static void
shim_stream_init(struct shim_stream *s)
{
    static const int t[SHIM_NARR] = { T_U8, T_I16, T_U8, T_U8, T_U8, T_I32, T_U8, T_U8, T_U8, T_U8, T_U8 };
    memset(s, 0, sizeof(*s));
    for (int i = 0; i < SHIM_NARR; ++i) s->arr[i].type = t[i];
    s->off.type = T_U32; s->mv_chk.type = T_I32; s->tmvp_exp.type = T_I32;
}

static void
shim_bind(OVCTUDec *c, int pic_w, int pic_h, uint8_t ict_type, uint8_t lmcs_flag)
{
    rcn_init_functions_hip(&c->rcn_funcs, ict_type, 1, 0, lmcs_flag, 10);
    ovhip_recorder *r = ovhip_rec_create(pic_w, pic_h);
    if (!r || ovhip_shim_bind_recorder(c, r, pic_w, pic_h)) { fprintf(stderr, "shim_bind failed\n"); exit(1); }
}

static size_t
shim_elem(int which)
{
    static const size_t e[SHIM_NARR] = { sizeof(ovhip_tb_cmd), 2, sizeof(ovhip_mc_unit), sizeof(ovhip_mc_unit), sizeof(ovhip_aff_unit), 4,
                                         sizeof(ovhip_lmcs_region), sizeof(ovhip_ciip_unit), sizeof(ovhip_dbf_edge), sizeof(ovhip_dbf_edge), sizeof(ovhip_itask) };
    return e[which];
}

/* one case done: take what the slots recorded, reset the recorder */
static void
shim_case_end(OVCTUDec *c, struct shim_stream *s, const char *what)
{
    ovhip_shim_flush_pending(c);
    if (ovhip_shim_last_error(c)) { fprintf(stderr, "shim: %s: slot latched error %d\n", what, ovhip_shim_last_error(c)); exit(1); }
    ovhip_recorder *r = ovhip_shim_recorder(c);
    uint32_t start[SHIM_NARR];
    const void *p[SHIM_NARR]; size_t n[SHIM_NARR];
    p[OVHIP_REC_TB] = ovhip_rec_tb_cmds(r, &n[OVHIP_REC_TB]);
    p[OVHIP_REC_COEF] = ovhip_rec_coefs(r, &n[OVHIP_REC_COEF]);
    p[OVHIP_REC_MC] = ovhip_rec_mc_units(r, &n[OVHIP_REC_MC]);
    p[OVHIP_REC_MCX] = ovhip_rec_mcx_units(r, &n[OVHIP_REC_MCX]);
    p[OVHIP_REC_AFF] = ovhip_rec_aff_units(r, &n[OVHIP_REC_AFF]);
    p[OVHIP_REC_SIDE] = ovhip_rec_aff_side(r, &n[OVHIP_REC_SIDE]);
    p[OVHIP_REC_REGION] = ovhip_rec_lmcs_regions(r, &n[OVHIP_REC_REGION]);
    p[OVHIP_REC_CIIP] = ovhip_rec_ciip_units(r, &n[OVHIP_REC_CIIP]);
    p[OVHIP_REC_EDGE_V] = ovhip_rec_dbf_edges(r, 0, &n[OVHIP_REC_EDGE_V], NULL);
    p[OVHIP_REC_EDGE_H] = ovhip_rec_dbf_edges(r, 1, &n[OVHIP_REC_EDGE_H], NULL);
    p[OVHIP_REC_ITASK] = ovhip_rec_itasks(r, &n[OVHIP_REC_ITASK]);
    for (int i = 0; i < SHIM_NARR; ++i) {
        const size_t es = shim_elem(i) / g_tsize[s->arr[i].type];
        start[i] = (uint32_t)(s->arr[i].n / es);
        if (n[i]) gbuf_push(&s->arr[i], p[i], n[i] * es);
    }
    gbuf_push(&s->off, start, SHIM_NARR);
    s->n_cases++;
    ovhip_rec_reset(r);
}

/* refs: the harness's reference pictures in the order the fixture stores them; the stream gets the slot -> picture map */
static void
shim_stream_write(const char *dir, const char *name, struct shim_stream *s, OVCTUDec *c, OVPicture **refs, int n_refs)
{
    static const char *nm[SHIM_NARR] = { "tb", "coef", "mc", "mcx", "aff", "side", "region", "ciip", "edge_v", "edge_h", "itask" };
    gfile g = gfile_open(dir, name);
    uint32_t end[SHIM_NARR];
    for (int i = 0; i < SHIM_NARR; ++i) {
        const size_t es = shim_elem(i) / g_tsize[s->arr[i].type];
        end[i] = (uint32_t)(s->arr[i].n / es);
        uint32_t d2[2] = { end[i], (uint32_t)es };
        gfile_array(&g, nm[i], s->arr[i].type, s->arr[i].data ? s->arr[i].data : (const void *)"", 2, d2);
    }
    gbuf_push(&s->off, end, SHIM_NARR);
    uint32_t d2[2] = { s->n_cases + 1, SHIM_NARR };
    gfile_array(&g, "case_off", T_U32, s->off.data, 2, d2);
    /* reference-picture table the slots built (order of first use) -> index into `refs` */
    extern int ovhip_shim_ref_pictures(const struct OVCTUDec *, const void **, int);
    const void *pics[16];
    int np = c ? ovhip_shim_ref_pictures(c, pics, 16) : 0;
    int32_t map[16];
    for (int i = 0; i < np; ++i) {
        map[i] = -1;
        for (int k = 0; k < n_refs; ++k) if ((const void *)refs[k] == pics[i]) map[i] = k;
    }
    uint32_t d1 = (uint32_t)np;
    gfile_array(&g, "ref_map", T_I32, np ? (const void *)map : (const void *)"", 1, &d1);
    if (s->mv_chk.n) gfile_buf(&g, "mv_patch_checked", &s->mv_chk);
    /* per plane entry the reference's flow writes: case, cell in the picture's plane, the unit whose vectors it holds */
    if (s->tmvp_exp.n) gfile_buf(&g, "tmvp_expected", &s->tmvp_exp);
    gfile_close(&g);
    fprintf(stderr, "%s: %u cases", name, s->n_cases);
    for (int i = 0; i < SHIM_NARR; ++i) if (end[i]) fprintf(stderr, ", %u %s", end[i], nm[i]);
    fprintf(stderr, "\n");
}
#endif
