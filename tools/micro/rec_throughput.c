/* Host-side throughput of the recorder on the call log of a real synthetic 4K picture (openvvc_amd/synth.py, calllog=True):
 * how many pictures per second ONE host thread can record -- every ovhip_rec_* call a parse thread makes for the picture
 * (ovhip_calllog_replay), then the class split the flush does -- and where the time goes by call type.
 *
 *   python -c "from openvvc_amd import synth; synth.make_workload(3840,2160,0x266,tools=synth.INTRA_TOOLS,intra_frac=0.12,calllog=True).calllog.tofile('/tmp/calllog_4k_b.bin')"
 *   gcc -O2 -I../../include rec_throughput.c -L../../openvvc_amd -lovvc_hip -Wl,-rpath,$PWD/../../openvvc_amd -o /tmp/rec_throughput
 *   /tmp/rec_throughput /tmp/calllog_4k_b.bin                                                                             */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "ovvc_hip.h"

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s calllog.bin [width height]\n", argv[0]); return 2; }
    const int W = argc > 3 ? atoi(argv[2]) : 3840, H = argc > 3 ? atoi(argv[3]) : 2160;
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    fseek(f, 0, SEEK_END);
    const size_t bytes = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    unsigned char *log = aligned_alloc(64, (bytes + 63) & ~(size_t)63);
    if (!log || fread(log, 1, bytes, f) != bytes) return 1;
    fclose(f);
    static const char *names[] = { "?", "ctu_size", "tu", "isp_cu", "pu", "affine_cu", "lmcs_region", "dbf_ctu", "ciip" };
    ovhip_recorder *rec = ovhip_rec_create(W, H);
    if (!rec) return 1;
    const int REPS = 20;
    /* whole picture */
    double t_all = 0, t_split = 0;
    int64_t calls = 0;
    for (int rep = 0; rep < REPS + 2; ++rep) {
        ovhip_rec_reset(rec);
        const double a = now();
        calls = ovhip_calllog_replay(log, bytes, rec);
        const double b = now();
        size_t counts[4], n; uint32_t nl; const uint32_t *ls; size_t nt;
        (void)ovhip_rec_tb_cmds_split(rec, counts, &n);
        (void)ovhip_rec_itasks_sorted(rec, &nt, &ls, &nl);
        const double c = now();
        if (calls < 0) { fprintf(stderr, "replay failed: %ld\n", (long)calls); return 1; }
        if (rep >= 2) { t_all += b - a; t_split += c - b; }
    }
    /* by call type: the records one at a time */
    double t_type[9] = { 0 }; long n_type[9] = { 0 };
    for (int rep = 0; rep < 5; ++rep) {
        ovhip_rec_reset(rec);
        for (size_t o = 0; o < bytes;) {
            uint32_t hdr[2];
            memcpy(hdr, log + o, 8);
            const double a = now();
            (void)ovhip_calllog_replay(log + o, 8 + hdr[1], rec);
            const double b = now();
            if (rep && hdr[0] < 9) { t_type[hdr[0]] += b - a; n_type[hdr[0]]++; }
            o += 8 + hdr[1];
        }
    }
    size_t n_tb, n_coef, n_mc, n_mcx;
    ovhip_rec_tb_cmds(rec, &n_tb); ovhip_rec_coefs(rec, &n_coef); ovhip_rec_mc_units(rec, &n_mc); ovhip_rec_mcx_units(rec, &n_mcx);
    const double ms = (t_all + t_split) / REPS * 1e3;
    printf("recorder, one %dx%d picture: %ld calls (%zu TB commands, %zu coefficients, %zu + %zu prediction units) in %.3f ms + %.3f ms class split / level sort "
           "-> %.0f pictures/s per host thread\n", W, H, (long)calls, n_tb, n_coef, n_mc, n_mcx, t_all / REPS * 1e3, t_split / REPS * 1e3, 1e3 / ms);
    for (int k = 1; k < 9; ++k)
        if (n_type[k]) printf("  %-12s %7ld calls/picture  %7.0f ns each  %6.3f ms/picture\n", names[k], n_type[k] / 4, t_type[k] / n_type[k] * 1e9, t_type[k] / 4 * 1e3);
    ovhip_rec_destroy(rec);
    return 0;
}
