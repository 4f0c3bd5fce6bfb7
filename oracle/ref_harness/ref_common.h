/* ref_common.h -- TEST INFRASTRUCTURE (this container only): helpers shared by the harnesses
 * that drive the compiled reference (oracle/_ref/libovvcref.so).  Our code; it includes the
 * reference's headers from where they lie (-I/root/reference/libovvc) and never ships.
 *
 * What the harness provides instead of the decoder around the rcn path:
 *   - a calloc'ed OVCTUDec whose function table is filled by the reference's own per-file
 *     initialisers (the body of rcn_init_functions(), libovvc/rcn.c:151-172, minus rcn.c
 *     itself, which needs the autoconf-generated ovconfig.h);
 *   - plain malloc'ed OVFrame / OVPicture objects with a no-op frame-sync function in slot 0
 *     (what dpb.c installs once a picture is complete, dpb.c:1236);
 *   - a fixture writer ("OVG1" container, read by tests/golden_io.py).
 */
#ifndef REF_COMMON_H
#define REF_COMMON_H

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#include "ovdefs.h"
#include "ovframe.h"
#include "ovdpb.h"
#include "dec_structures.h"
#include "ctudec.h"
#include "rcn_structures.h"
#include "rcn.h"
#include "drv.h"
#include "drv_utils.h"

/* ------------------------------------------------------------------ seeded LCG */
static uint32_t g_seed = 0x266;
static inline uint32_t rnd32(void) { g_seed = g_seed * 1664525u + 1013904223u; return g_seed >> 8; }
static inline int rnd_range(int lo, int hi) { return lo + (int)(rnd32() % (uint32_t)(hi - lo + 1)); }

/* ------------------------------------------------------------------ function table */
void rcn_init_ctu_buffs_10(struct RCNFunctions *);
void rcn_init_mc_functions_10(struct RCNFunctions *);
void rcn_init_tr_functions_10(struct RCNFunctions *);
void rcn_init_dc_planar_functions_10(struct RCNFunctions *);
void rcn_init_ict_functions_10(struct RCNFunctions *, uint8_t type, uint8_t bitdepth);
void rcn_init_lfnst_functions(struct RCNFunctions *);
void rcn_init_mip_functions_10(struct RCNFunctions *);
void rcn_init_alf_functions_10(struct RCNFunctions *);
void rcn_init_sao_functions_10(struct RCNFunctions *);
void rcn_init_lmcs_function_10(struct RCNFunctions *, uint8_t lmcs_flag);
void rcn_init_dmvr_functions_10(struct RCNFunctions *);
void rcn_init_prof_functions_10(struct RCNFunctions *);
void rcn_init_bdof_functions_10(struct RCNFunctions *);
void rcn_init_ciip_functions_10(struct RCNFunctions *);
void rcn_init_df_functions_10(struct RCNFunctions *);
void rcn_init_dequant_10(struct RCNFunctions *);
void rcn_init_fill_ref_10(struct RCNFunctions *);
void rcn_init_transform_trees_10(struct RCNFunctions *);
void rcn_init_intra_functions_10(struct RCNFunctions *);
void rcn_init_inter_functions_10(struct RCNFunctions *);
void rcn_init_ibc_10(struct RCNFunctions *);
void rcn_init_cclm_functions_10(struct RCNFunctions *);

/* the initialisers of the reference's x86 back-end (prototypes as in libovvc/x86/rcn_sse.h:39-54, rcn_avx2.h:38-47; those headers
 * include the autoconf-generated ovconfig.h and cannot be included here) */
void rcn_init_mc_functions_sse(struct RCNFunctions *const);
void rcn_init_tr_functions_sse(struct RCNFunctions *const);
void rcn_init_dc_planar_functions_sse(struct RCNFunctions *const);
void rcn_init_ict_functions_sse(struct RCNFunctions *, uint8_t type);
void rcn_init_alf_functions_sse(struct RCNFunctions *);
void rcn_init_cclm_functions_sse(struct RCNFunctions *);
void rcn_init_lfnst_functions_sse(struct RCNFunctions *);
void rcn_init_mip_functions_sse(struct RCNFunctions *const);
void rcn_init_dmvr_functions_sse(struct RCNFunctions *const);
void rcn_init_prof_functions_sse(struct RCNFunctions *const);
void rcn_init_bdof_functions_sse(struct RCNFunctions *const);
void rcn_init_ciip_functions_sse(struct RCNFunctions *const);
void rcn_init_df_functions_sse(struct RCNFunctions *const);
void rcn_init_intra_angular_functions_10_sse(struct RCNFunctions *);
void rcn_init_dequant_sse(struct RCNFunctions *);
void rcn_init_alf_functions_avx2(struct RCNFunctions *);
void rcn_init_sao_functions_avx2(struct RCNFunctions *);
void rcn_init_ict_functions_avx2(struct RCNFunctions *, uint8_t type);
void rcn_init_mip_functions_avx2(struct RCNFunctions *const);
void rcn_init_prof_functions_avx2(struct RCNFunctions *const);
void rcn_init_bdof_functions_avx2(struct RCNFunctions *const);
void rcn_init_dmvr_functions_avx2(struct RCNFunctions *const);
void rcn_init_ciip_functions_avx2(struct RCNFunctions *const);
void rcn_init_mc_functions_avx2(struct RCNFunctions *const);
void rcn_init_intra_angular_functions_10_avx2(struct RCNFunctions *);
/* != 0: the reference's SSE4.1 / AVX2 overrides on top of the scalar table, in the order of rcn.c:216-254 (all but
 * rcn_init_sao_functions_sse, whose file needs the autoconf-generated ovconfig.h and is not built here) */
static int g_simd;

/* same call sequence as rcn_init_functions() (rcn.c:151-180) for bitdepth 10, scalar */
static void ref_fill_table_scalar(struct RCNFunctions *f, uint8_t ict_type, uint8_t lmcs_flag);
static void
ref_fill_table(struct RCNFunctions *f, uint8_t ict_type, uint8_t lmcs_flag)
{
    ref_fill_table_scalar(f, ict_type, lmcs_flag);
    if (!g_simd) return;
    if (!__builtin_cpu_supports("sse4.1") || !__builtin_cpu_supports("avx2")) { fprintf(stderr, "simd: this CPU has no SSE4.1 / AVX2\n"); exit(1); }
    rcn_init_mc_functions_sse(f);
    rcn_init_tr_functions_sse(f);
    rcn_init_dc_planar_functions_sse(f);
    rcn_init_ict_functions_sse(f, ict_type);
    rcn_init_lfnst_functions_sse(f);
    rcn_init_mip_functions_sse(f);
    rcn_init_alf_functions_sse(f);
    rcn_init_dmvr_functions_sse(f);
    rcn_init_prof_functions_sse(f);
    rcn_init_bdof_functions_sse(f);
    rcn_init_ciip_functions_sse(f);
    rcn_init_df_functions_sse(f);
    rcn_init_intra_angular_functions_10_sse(f);
    rcn_init_dequant_sse(f);
    rcn_init_cclm_functions_sse(f);                  /* lm_chroma_enabled && !sps_chroma_vertical_collocated_flag (rcn.c:233-237) */
    rcn_init_alf_functions_avx2(f);
    rcn_init_sao_functions_avx2(f);
    rcn_init_ict_functions_avx2(f, ict_type);
    rcn_init_mip_functions_avx2(f);
    rcn_init_ciip_functions_avx2(f);
    rcn_init_mc_functions_avx2(f);
    rcn_init_dmvr_functions_avx2(f);
    rcn_init_prof_functions_avx2(f);
    rcn_init_bdof_functions_avx2(f);
    rcn_init_intra_angular_functions_10_avx2(f);
}

static void
ref_fill_table_scalar(struct RCNFunctions *f, uint8_t ict_type, uint8_t lmcs_flag)
{
    rcn_init_ctu_buffs_10(f);
    rcn_init_mc_functions_10(f);
    rcn_init_tr_functions_10(f);
    rcn_init_dc_planar_functions_10(f);
    rcn_init_ict_functions_10(f, ict_type, 10);
    {   /* rcn_init_intra_angular_functions(rcn_func, 10) is static in rcn.c (:69-110): the nine exported function tables */
        extern const struct IntraAngularFunctions angular_gauss_h_10, angular_gauss_v_10, angular_cubic_h_10, angular_cubic_v_10,
                                                   angular_c_h_10, angular_c_v_10, angular_nofrac_v_10, angular_nofrac_h_10;
        extern const struct IntraMRLFunctions mrl_func_10;
        f->intra_angular_gauss_h = &angular_gauss_h_10; f->intra_angular_gauss_v = &angular_gauss_v_10;
        f->intra_angular_cubic_h = &angular_cubic_h_10; f->intra_angular_cubic_v = &angular_cubic_v_10;
        f->intra_angular_c_h = &angular_c_h_10; f->intra_angular_c_v = &angular_c_v_10;
        f->intra_angular_nofrac_v = &angular_nofrac_v_10; f->intra_angular_nofrac_h = &angular_nofrac_h_10;
        f->intra_mrl = &mrl_func_10;
    }
    rcn_init_lfnst_functions(f);
    rcn_init_mip_functions_10(f);
    rcn_init_alf_functions_10(f);
    rcn_init_sao_functions_10(f);
    rcn_init_lmcs_function_10(f, lmcs_flag);
    rcn_init_dmvr_functions_10(f);
    rcn_init_prof_functions_10(f);
    rcn_init_bdof_functions_10(f);
    rcn_init_ciip_functions_10(f);
    rcn_init_df_functions_10(f);
    rcn_init_dequant_10(f);
    rcn_init_fill_ref_10(f);
    rcn_init_transform_trees_10(f);
    rcn_init_intra_functions_10(f);
    rcn_init_inter_functions_10(f);
    rcn_init_ibc_10(f);
    rcn_init_cclm_functions_10(f);
}

/* ------------------------------------------------------------------ fake decoder state */
static OVPartInfo g_part = { .log2_ctu_s = 7, .log2_min_cb_s = 2 };

static OVCTUDec *
ref_new_ctudec(uint8_t ict_type, uint8_t lmcs_flag)
{
    OVCTUDec *c = NULL;
    if (posix_memalign((void **)&c, 64, sizeof(*c))) abort();
    memset(c, 0, sizeof(*c));
    ref_fill_table(&c->rcn_funcs, ict_type, lmcs_flag);
    c->rcn_ctx.ctudec = c;
    c->part_ctx = &g_part;
    c->part_ctx_c = &g_part;
    c->rcn_funcs.rcn_attach_ctu_buff(&c->rcn_ctx, 7, 0);
    return c;
}

static void ref_no_synchro(const OVPicture *const p, int a, int b, int c, int d) { (void)p; (void)a; (void)b; (void)c; (void)d; }

static OVPicture *
ref_new_picture(int w, int h, int poc)
{
    OVPicture *p = calloc(1, sizeof(*p));
    OVFrame *f = calloc(1, sizeof(*f));
    f->width = w; f->height = h;
    f->linesize[0] = (size_t)w * 2; f->linesize[1] = f->linesize[2] = (size_t)(w / 2) * 2;
    f->data[0] = calloc((size_t)w * h, 2);
    f->data[1] = calloc((size_t)(w / 2) * (h / 2), 2);
    f->data[2] = calloc((size_t)(w / 2) * (h / 2), 2);
    p->frame = f;
    p->poc = poc;
    atomic_init(&p->idx_function, 0);
    p->ovdpb_frame_synchro[0] = ref_no_synchro;
    p->ovdpb_frame_synchro[1] = ref_no_synchro;
    return p;
}

/* smooth-ish random 10-bit content: random walk + noise, so filters/decisions are non-degenerate */
static void
fill_plane(uint16_t *p, int w, int h, int stride)
{
    for (int y = 0; y < h; ++y) {
        int v = rnd_range(100, 900);
        for (int x = 0; x < w; ++x) {
            v += rnd_range(-24, 24);
            if (y) v = (v + p[(y - 1) * stride + x]) >> 1;
            if (rnd_range(0, 31) == 0) v = rnd_range(0, 1023);
            if (v < 0) v = 0;
            if (v > 1023) v = 1023;
            p[y * stride + x] = (uint16_t)v;
        }
    }
}

/* ------------------------------------------------------------------ fixture container "OVG1" */
enum { T_U8 = 0, T_I16 = 1, T_U16 = 2, T_I32 = 3, T_U32 = 4, T_U64 = 5, T_I8 = 6 };
static const int g_tsize[] = { 1, 2, 2, 4, 4, 8, 1 };

typedef struct { void *data; size_t n, cap; int type; } gbuf;

static void
gbuf_push(gbuf *b, const void *src, size_t n_elem)
{
    size_t es = g_tsize[b->type];
    if ((b->n + n_elem) * es > b->cap) {
        size_t nc = b->cap ? b->cap : 4096;
        while (nc < (b->n + n_elem) * es) nc *= 2;
        b->data = realloc(b->data, nc);
        b->cap = nc;
    }
    memcpy((char *)b->data + b->n * es, src, n_elem * es);
    b->n += n_elem;
}

typedef struct { FILE *f; uint32_t n; } gfile;

static gfile
gfile_open(const char *dir, const char *name)
{
    char path[1024];
    gfile g;
    snprintf(path, sizeof(path), "%s/%s", dir, name);
    g.f = fopen(path, "wb");
    if (!g.f) { perror(path); exit(1); }
    g.n = 0;
    uint32_t hdr[2] = { 0x3147564f /* "OVG1" */, 0 };
    fwrite(hdr, 4, 2, g.f);
    return g;
}

static void
gfile_array(gfile *g, const char *name, int type, const void *data, int ndim, const uint32_t *dims)
{
    char nm[32] = { 0 };
    uint32_t hdr[2 + 4] = { (uint32_t)type, (uint32_t)ndim, 1, 1, 1, 1 };
    size_t total = 1;
    strncpy(nm, name, 31);
    for (int i = 0; i < ndim; ++i) { hdr[2 + i] = dims[i]; total *= dims[i]; }
    fwrite(nm, 1, 32, g->f);
    fwrite(hdr, 4, 6, g->f);
    fwrite(data, g_tsize[type], total, g->f);
    g->n++;
}

static void gfile_buf(gfile *g, const char *name, const gbuf *b)
{
    uint32_t d = (uint32_t)b->n;
    gfile_array(g, name, b->type, b->data ? b->data : (const void *)"", 1, &d);
}

static void
gfile_close(gfile *g)
{
    fseek(g->f, 4, SEEK_SET);
    fwrite(&g->n, 4, 1, g->f);
    fclose(g->f);
}

#endif
