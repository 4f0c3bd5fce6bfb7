/* Host-side throughput of the recorder (ovhip_rec_pu / ovhip_rec_tu / ovhip_rec_tb_cmds_split) on a synthetic 4K
 * picture's worth of descriptors (same counts and mixes as openvvc_amd/synth.py): how many pictures per second one
 * host thread can record.  Build:  gcc -O2 -I../../include rec_throughput.c -L../../openvvc_amd -lovvc_hip \
 *                                      -Wl,-rpath,$PWD/../../openvvc_amd -o /tmp/rec_throughput                     */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "ovvc_hip.h"

static uint32_t rng = 0x266;
static uint32_t rnd(void) { rng ^= rng << 13; rng ^= rng >> 17; rng ^= rng << 5; return rng; }
static double frand(void) { return (rnd() >> 8) * (1.0 / 16777216.0); }
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(void)
{
    enum { W = 3840, H = 2160, NPU = 28242, NTU = 22775 };
    static ovhip_pu_desc pus[NPU];
    static ovhip_tu_desc tus[NTU];
    static int16_t pool[3][64 * 64];
    for (int c = 0; c < 3; ++c) for (int i = 0; i < 64 * 64; ++i) pool[c][i] = (int16_t)((int)(rnd() % 61) - 30);
    for (int i = 0; i < NPU; ++i) {
        ovhip_pu_desc *p = &pus[i];
        memset(p, 0, sizeof(*p));
        p->log2_w = 3 + rnd() % 4; p->log2_h = 3 + rnd() % 4;
        p->x0 = (rnd() % ((W >> p->log2_w))) << p->log2_w; p->y0 = (rnd() % ((H >> p->log2_h))) << p->log2_h;
        p->inter_dir = frand() < 0.6 ? 3 : 1 + rnd() % 2;
        p->planes = 3; p->lmcs = 1;
        p->mv0x = (int)(rnd() % 2048) - 1024; p->mv0y = (int)(rnd() % 2048) - 1024;
        p->mv1x = -p->mv0x + (int)(rnd() % 64) - 32; p->mv1y = -p->mv0y + (int)(rnd() % 64) - 32;
        p->poc0 = 0; p->poc1 = 16; p->ref0 = 0; p->ref1 = 1;
        p->bcw_idx_plus1 = frand() < 0.1 ? 1 + rnd() % 5 : 0;
        if (p->inter_dir == 3 && !p->bcw_idx_plus1 && p->log2_h >= 3 && p->log2_w + p->log2_h >= 7) {
            const double u = frand();
            p->refine = u < 0.45 ? OVHIP_PU_BDOF : u < 0.8 ? (OVHIP_PU_DMVR | OVHIP_PU_BDOF) : 0;
        }
    }
    for (int i = 0; i < NTU; ++i) {
        ovhip_tu_desc *t = &tus[i];
        memset(t, 0, sizeof(*t));
        t->log2_tb_w = 3 + rnd() % 4; t->log2_tb_h = 3 + rnd() % 4;
        t->x0 = (rnd() % ((W >> t->log2_tb_w))) << t->log2_tb_w; t->y0 = (rnd() % ((H >> t->log2_tb_h))) << t->log2_tb_h;
        t->tree = 0;
        t->cbf_mask = 0x10 | (frand() < 0.5 ? 0x2 : 0) | (frand() < 0.5 ? 0x1 : 0);
        for (int c = 0; c < 3; ++c) {
            const int l2w = c == 2 ? t->log2_tb_w : t->log2_tb_w - 1, l2h = c == 2 ? t->log2_tb_h : t->log2_tb_h - 1;
            const int nx = (l2w > 5 ? 32 : 1 << l2w) / 4, ny = (l2h > 5 ? 32 : 1 << l2h) / 4;
            uint64_t m = 0;
            const int lx = 1 + rnd() % (nx < 3 ? nx : 3), ly = 1 + rnd() % (ny < 3 ? ny : 3);
            for (int sy = 0; sy < ly; ++sy) for (int sx = 0; sx < lx; ++sx) if (!(sx | sy) || frand() < 0.65) m |= 1ull << (sy * 8 + sx);
            t->sig_sb_map[c] = frand() < 0.25 ? 0 : m;
            t->last_pos[c] = t->sig_sb_map[c] ? 0x0101 : 0;
            t->coef[c] = pool[c];
        }
        pool[0][0] = pool[1][0] = pool[2][0] = 3;
    }
    ovhip_tu_state st;
    memset(&st, 0, sizeof(st));
    st.qp_y = 32; st.qp_cb = st.qp_cr = 31; st.qp_jcbcr = 30; st.qp_y_skip = 32; st.qp_cb_skip = st.qp_cr_skip = st.qp_jcbcr_skip = 31;
    st.dep_quant = 1;
    ovhip_recorder *rec = ovhip_rec_create(W, H);
    if (!rec) return 1;
    double t_pu = 0, t_tu = 0, t_split = 0;
    const int REPS = 20;
    long bad = 0;
    for (int rep = 0; rep < REPS + 1; ++rep) {
        ovhip_rec_reset(rec);
        double a = now();
        for (int i = 0; i < NPU; ++i) bad += ovhip_rec_pu(rec, &pus[i]) < 0;
        double b = now();
        for (int i = 0; i < NTU; ++i) bad += ovhip_rec_tu(rec, &st, &tus[i]) < 0;
        double c = now();
        size_t counts[4], n;
        (void)ovhip_rec_tb_cmds_split(rec, counts, &n);
        double d = now();
        if (rep) { t_pu += b - a; t_tu += c - b; t_split += d - c; }
    }
    const double ms = (t_pu + t_tu + t_split) / REPS * 1e3;
    printf("recorder, one 4K picture: %d PUs %.0f ns each, %d TUs %.0f ns each, class split %.2f ms -> %.2f ms per picture, "
           "%.0f pictures/s per host thread (rejected descriptors: %ld)\n",
           NPU, t_pu / REPS / NPU * 1e9, NTU, t_tu / REPS / NTU * 1e9, t_split / REPS * 1e3, ms, 1e3 / ms, bad / (REPS + 1));
    ovhip_rec_destroy(rec);
    return 0;
}
