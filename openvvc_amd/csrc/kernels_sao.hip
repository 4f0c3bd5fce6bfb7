// kernels_sao.hip -- K13 on gfx950: sample adaptive offset (band / edge), all three planes in one
// launch.  Source = deblocked picture, destination = a second picture (ping-pong): the
// frame-resident form of the reference's filter_region copies with saved rows/columns
// (libovvc/rcn_ctu.c:246-510).  Pure streaming op: each lane owns 8 consecutive samples of a row
// (one 16-byte load of the centre, neighbours from L1), parameters of the containing CTU are read
// through the scalar cache.  Replaces sao.band / sao.edge and the drivers rcn_sao_ctu,
// rcn_sao_filter_line, rcn_sao_first_pix_rows (libovvc/rcn_sao.c:46-293).
#include "ovvc_common.hip.h"

namespace {

__global__ __launch_bounds__(256) void k_sao(ovhip_pic dst, ovhip_pic src, const ovhip_sao_ctu *__restrict__ prm,
                                              int log2_ctu, int nb_ctu_w)
{
    const int c = blockIdx.z;
    const int sh = c ? 1 : 0;
    const int w = src.w >> sh, h = src.h >> sh, l2 = log2_ctu - sh;
    const int y = blockIdx.y;
    if (y >= h) return;
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (x0 >= w) return;
    int ss, ds;
    const uint16_t *s = ov_plane(src, c, ss) + y * ss;
    uint16_t *d = ov_plane(dst, c, ds) + y * ds;
    const ovhip_sao_ctu p = prm[(y >> l2) * nb_ctu_w + (x0 >> l2)];   // 8 | ctu size: one CTU per lane
    const int type = p.type[c];
    const int n = min(8, w - x0);
    int v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = i < n ? s[x0 + i] : 0;
    int o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = v[i];
    if (type == OVHIP_SAO_BAND) {
        const int bp = p.band_position[c];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = ((v[i] >> (OV_BD - 5)) - bp) & 31;
            if (k < 4) o[i] = ov_clip_bd(v[i] + p.offset_val[c][k]);
        }
    } else if (type == OVHIP_SAO_EDGE) {
        const int eo = p.eo_class[c];
        const int dxa = eo == 1 ? 0 : (eo == 3 ? 1 : -1), dya = eo == 0 ? 0 : -1;
        // last term: quirk of the reference for pictures of a single CTU row (rcn_sao.c:262): the first
        // 6-row band is processed with the BOTTOM border flag, its last row is skipped
        const bool rowskip = eo != 0 && (y == 0 || y == h - 1 || (src.h <= (1 << log2_ctu) && y == (6 >> sh) - 1));
        if (!rowskip) {
            const uint16_t *sa = s + dya * ss, *sb = s - dya * ss;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int x = x0 + i;
                if (i < n && !(eo != 1 && (x == 0 || x == w - 1))) {
                    const int a = sa[x + dxa], b = sb[x - dxa];
                    const int idx = 2 + (v[i] > a) - (v[i] < a) + (v[i] > b) - (v[i] < b);
                    o[i] = ov_clip_bd(v[i] + p.offset_val[c][idx]);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) if (i < n) d[x0 + i] = (uint16_t)o[i];
}

} // namespace

extern "C" int ovhip_sao_launch(ovhip_ctx *ctx, const ovhip_pic *dst, const ovhip_pic *src,
                                const ovhip_sao_ctu *d_params, int32_t log2_ctu_s)
{
    if (!ctx || !dst || !src || !d_params) return OVHIP_EINVAL;
    if (dst->w != src->w || dst->h != src->h || dst->y == src->y || log2_ctu_s < 5 || log2_ctu_s > 7)
        return ov_fail(ctx, OVHIP_EINVAL, "ovhip_sao_launch: bad pictures / CTU size", hipSuccess);
    const int nb_ctu_w = (src->w + (1 << log2_ctu_s) - 1) >> log2_ctu_s;
    dim3 grid((src->w / 8 + 255) / 256 + 1, src->h, 3);
    hipLaunchKernelGGL(k_sao, grid, dim3(256), 0, ctx->stream, *dst, *src, d_params, log2_ctu_s, nb_ctu_w);
    OV_LAUNCH_CHECK(ctx, "k_sao");
    return OVHIP_OK;
}
