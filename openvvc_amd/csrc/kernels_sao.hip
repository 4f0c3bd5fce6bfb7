// kernels_sao.hip -- K13 on gfx950: sample adaptive offset (band / edge), all three planes in one
// launch.  Source = deblocked picture, destination = a second picture (ping-pong): the
// frame-resident form of the reference's filter_region copies with saved rows/columns
// (libovvc/rcn_ctu.c:246-510).  Pure streaming op: each lane owns 8 consecutive samples of a row
// (one 16-byte load of the centre, neighbours from L1), parameters of the containing CTU are read
// through the scalar cache.  Replaces sao.band / sao.edge and the drivers rcn_sao_ctu,
// rcn_sao_filter_line, rcn_sao_first_pix_rows (libovvc/rcn_sao.c:46-293).
#include "ovvc_common.hip.h"
#include <stdlib.h>

namespace {

#define SAO_TW 64      /* tile: 64 x 32 samples = 256 lanes x 8 samples; never straddles a CTU (luma 128, chroma 64) */
#define SAO_TH 32

struct Vec8 { int v[8]; };

__device__ __forceinline__ void load8(const uint16_t *p, bool vec, int n, int v[8])
{
    if (vec) {
        const uint4 q = *reinterpret_cast<const uint4 *>(p);
        v[0] = q.x & 0xffff; v[1] = q.x >> 16; v[2] = q.y & 0xffff; v[3] = q.y >> 16;
        v[4] = q.z & 0xffff; v[5] = q.z >> 16; v[6] = q.w & 0xffff; v[7] = q.w >> 16;
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = i < n ? p[i] : 0;
    }
}

__global__ __launch_bounds__(256) void k_sao(ovhip_pic dst, ovhip_pic src, const ovhip_sao_ctu *__restrict__ prm,
                                              int log2_ctu, int nb_ctu_w, int tiles_y, int tiles_c, int ty0_y, int ty0_c, int row0, int row1)
{
    const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;          // 8 lanes x 8 samples per row, 32 rows
    // (plane, tile): luma tiles first, then Cb, then Cr; one tile per workgroup (single pass: `continue` leaves the tile)
    for (int t = blockIdx.x; t < tiles_y + 2 * tiles_c; t = tiles_y + 2 * tiles_c) {
        const int c = t < tiles_y ? 0 : (t < tiles_y + tiles_c ? 1 : 2);
        // XCD-aware tile order (ov_xcd_slot_at): an XCD takes one contiguous band of each plane, the band k_alf's workgroups on the same
        // XCD read next; the rows above and below a tile come out of the L2 that fetched them for the neighbouring tile
        const int tt = (int)ov_xcd_slot_at(t, c == 0 ? 0 : (c == 1 ? tiles_y : tiles_y + tiles_c), c == 0 ? tiles_y : tiles_c);
        const int sh = c ? 1 : 0;
        const int w = src.w >> sh, h = src.h >> sh, l2 = log2_ctu - sh;
        const int ntx = (w + SAO_TW - 1) / SAO_TW;
        // (ty0_*: first tile row of the launch's row window, ovhip_sao_launch_rows; 0 for a whole picture)
        const int x0 = (tt % ntx) * SAO_TW + tx * 8, y = (tt / ntx + (c ? ty0_c : ty0_y)) * SAO_TH + ty;
        if (x0 >= w || y >= h || y < (row0 >> sh) || y >= (row1 >> sh)) continue;      // (row0, row1: the launch's row window, luma rows)
        int ss, ds;
        const uint16_t *s = ov_plane(src, c, ss) + y * ss;
        uint16_t *d = ov_plane(dst, c, ds) + y * ds;
        const ovhip_sao_ctu *p = &prm[(y >> l2) * nb_ctu_w + (x0 >> l2)];   // tile-uniform -> scalar loads
        const int type = p->type[c];
        const int n = min(8, w - x0);
        const bool vec = n == 8 && !((ss | ds) & 7) && !(((uintptr_t)s | (uintptr_t)d) & 15);
        int v[8], o[8];
        load8(s + x0, vec, n, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = v[i];
        if (type == OVHIP_SAO_BAND) {
            const int bp = p->band_position[c];
            const int o0 = p->offset_val[c][0], o1 = p->offset_val[c][1], o2 = p->offset_val[c][2], o3 = p->offset_val[c][3];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = ((v[i] >> (OV_BD - 5)) - bp) & 31;
                const int off = k == 0 ? o0 : k == 1 ? o1 : k == 2 ? o2 : k == 3 ? o3 : 0;
                o[i] = ov_clip_bd(v[i] + off);
            }
        } else if (type == OVHIP_SAO_EDGE) {
            const int eo = p->eo_class[c];
            const int dxa = eo == 1 ? 0 : (eo == 3 ? 1 : -1), dya = eo == 0 ? 0 : -1;
            // last term: quirk of the reference for pictures of a single CTU row (rcn_sao.c:262): the first
            // 6-row band is processed with the BOTTOM border flag, its last row is skipped
            // borders of the CTU's rect entry (tile) inside the picture count like the picture's (is_border from the entry-local CTU
            // index, rcn_sao.c:211-214, :253-257): ovhip_sao_ctu.border, zero in a picture of one entry
            const int bd = p->border, cs = 1 << l2;
            const int cy0 = y & ~(cs - 1), cx0 = x0 & ~(cs - 1);
            const int cx1 = min(cx0 + cs, w) - 1;
            const bool rowskip = eo != 0 && (y == 0 || y == h - 1 || (src.h <= (1 << log2_ctu) && y == (6 >> sh) - 1) ||
                                             ((bd & OVHIP_BORDER_UPPER) && y == cy0) || ((bd & OVHIP_BORDER_BOTTOM) && y == min(cy0 + cs, h) - 1) ||
                                             ((bd & OVHIP_BORDER_ONE_ROW) && y == cy0 + (6 >> sh) - 1));
            const int skip_l = (bd & OVHIP_BORDER_LEFT) ? cx0 : -1, skip_r = (bd & OVHIP_BORDER_RIGHT) ? cx1 : -1;
            if (!rowskip) {
                const int of0 = p->offset_val[c][0], of1 = p->offset_val[c][1], of2 = p->offset_val[c][2],
                          of3 = p->offset_val[c][3], of4 = p->offset_val[c][4];
                // neighbour rows: a = (x + dxa, y + dya), b = (x - dxa, y - dya); 8 centre-aligned samples plus
                // one extra on each side cover every class
                int ra[10], rb[10];
                const uint16_t *sa = s + dya * ss, *sb = s - dya * ss;
                load8(sa + x0, vec, n, ra + 1);
                load8(sb + x0, vec, n, rb + 1);
                ra[0] = x0 > 0 ? sa[x0 - 1] : 0; rb[0] = x0 > 0 ? sb[x0 - 1] : 0;
                ra[9] = x0 + 8 < w ? sa[x0 + 8] : 0; rb[9] = x0 + 8 < w ? sb[x0 + 8] : 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int x = x0 + i;
                    if (i < n && !(eo != 1 && (x == 0 || x == w - 1 || x == skip_l || x == skip_r))) {
                        const int a = ra[1 + i + dxa], b = rb[1 + i - dxa];
                        const int idx = 2 + (v[i] > a) - (v[i] < a) + (v[i] > b) - (v[i] < b);
                        const int off = idx == 0 ? of0 : idx == 1 ? of1 : idx == 2 ? of2 : idx == 3 ? of3 : of4;
                        o[i] = ov_clip_bd(v[i] + off);
                    }
                }
            }
        }
        if (vec) {
            uint4 q;
            q.x = o[0] | (o[1] << 16); q.y = o[2] | (o[3] << 16); q.z = o[4] | (o[5] << 16); q.w = o[6] | (o[7] << 16);
            *reinterpret_cast<uint4 *>(d + x0) = q;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) if (i < n) d[x0 + i] = (uint16_t)o[i];
        }
    }
}

} // namespace

// Rows [row0, row1) of the picture (luma rows; both multiples of 8, or row1 = the picture's height): what the band-wise picture job
// runs behind the deblocking of a band (ovvc_picture.hip).  Reads src rows row0 - 1 .. row1 (the edge classes' neighbours), writes dst
// rows [row0, row1) only -- the lanes of a tile's rows outside the window leave at once.  The samples are computed exactly as by the
// whole-picture launch.
extern "C" int ovhip_sao_launch_rows(ovhip_ctx *ctx, const ovhip_pic *dst, const ovhip_pic *src,
                                     const ovhip_sao_ctu *d_params, int32_t log2_ctu_s, int32_t row0, int32_t row1)
{
    if (!ctx || !dst || !src || !d_params) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    if (dst->w != src->w || dst->h != src->h || dst->y == src->y || log2_ctu_s < 5 || log2_ctu_s > 7)
        return ov_fail(ctx, OVHIP_EINVAL, "ovhip_sao_launch: bad pictures / CTU size", hipSuccess);
    if (row0 < 0 || row1 > src->h || row0 > row1 || (row0 & 7) || ((row1 & 7) && row1 != src->h))
        return ov_fail(ctx, OVHIP_EINVAL, "ovhip_sao_launch_rows: row window", hipSuccess);
    if (row0 == row1) return OVHIP_OK;
    const int nb_ctu_w = (src->w + (1 << log2_ctu_s) - 1) >> log2_ctu_s;
    const int ty0_y = row0 / SAO_TH, ty1_y = (row1 + SAO_TH - 1) / SAO_TH;
    const int ty0_c = (row0 / 2) / SAO_TH, ty1_c = (row1 / 2 + SAO_TH - 1) / SAO_TH;
    const int tiles_y = ((src->w + SAO_TW - 1) / SAO_TW) * (ty1_y - ty0_y);
    const int tiles_c = ((src->w / 2 + SAO_TW - 1) / SAO_TW) * (ty1_c - ty0_c);
    const int total = tiles_y + 2 * tiles_c;
    // one workgroup per tile (measured: 18.7 us; a resident grid of 2048 workgroups 22.4 us, 1024: 26.5 us)
    hipLaunchKernelGGL(k_sao, dim3(total), dim3(256), 0, ctx->stream, *dst, *src, d_params,
                       log2_ctu_s, nb_ctu_w, tiles_y, tiles_c, ty0_y, ty0_c, row0, row1);
    OV_LAUNCH_CHECK(ctx, "k_sao");
    return OVHIP_OK;
}

extern "C" int ovhip_sao_launch(ovhip_ctx *ctx, const ovhip_pic *dst, const ovhip_pic *src,
                                const ovhip_sao_ctu *d_params, int32_t log2_ctu_s)
{
    if (!src) return OVHIP_EINVAL;
    return ovhip_sao_launch_rows(ctx, dst, src, d_params, log2_ctu_s, 0, src->h);
}
