// TMVP motion plane: where the vectors DMVR refined belong in the picture's collocated motion plane -- the
// address arithmetic of the reference's caller (vcl_coding_unit.c:2629-2645) and of tmvp_store_mv (drv_lines.c:270-330) on the
// device, see include/ovvc_hip.h "TMVP motion plane".
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ovvc_hip.h"
#include "ovvc_common.hip.h"

namespace {

__global__ __launch_bounds__(256) void k_tmvp_cells(const ovhip_mc_unit *__restrict__ units, uint32_t n, const int32_t *__restrict__ refined,
                                                    int log2_ctu, int nb_ctb_w, ovhip_tmvp_cell *__restrict__ out)
{
    const uint32_t u = blockIdx.x * 256 + threadIdx.x;
    if (u >= n) return;
    const ovhip_mc_unit t = units[u];
    ovhip_tmvp_cell c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { c[k].cell = OVHIP_TMVP_NONE; c[k].mv0x = c[k].mv0y = c[k].mv1x = c[k].mv1y = 0; }
    if (t.flags & OVHIP_MC_DMVR) {
        const int ctu = 1 << log2_ctu, nb = ctu >> 3, stride = nb * nb_ctb_w;
        const int x0 = t.x & (ctu - 1), y0 = t.y & (ctu - 1);
        const int ux = (x0 + 7) >> 3, uy = (y0 + 7) >> 3;                   // cell of the block in the CTU's array
        const int cx = (t.x >> log2_ctu) * nb, cy = (t.y >> log2_ctu) * nb; // the CTU's first cell in the plane
        const int32_t *m = refined + 4 * (size_t)u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int dx = k & 1, dy = k >> 1;
            if ((dx && t.w <= 8) || (dy && t.h <= 8)) continue;            // log2_w > 3 / log2_h > 3
            if (ux + dx >= nb || uy + dy >= nb) continue;                   // outside the rows / columns tmvp_store_mv copies
            c[k].cell = (uint32_t)((cy + uy + dy) * stride + cx + ux + dx);
            c[k].mv0x = m[0]; c[k].mv0y = m[1]; c[k].mv1x = m[2]; c[k].mv1y = m[3];
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) out[4 * (size_t)u + k] = c[k];
}

} // namespace

extern "C" int ovhip_tmvp_cells_launch(ovhip_ctx *ctx, const ovhip_mc_unit *d_units, uint32_t n_units, const int32_t *d_refined,
                                       int32_t log2_ctu_s, int32_t nb_ctb_w, ovhip_tmvp_cell *d_out)
{
    if (!ctx) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    if (!n_units) return OVHIP_OK;
    if (!d_units || !d_refined || !d_out || log2_ctu_s < 5 || log2_ctu_s > 7 || nb_ctb_w <= 0)
        return ov_fail(ctx, OVHIP_EINVAL, "ovhip_tmvp_cells_launch: bad arguments", hipSuccess);
    hipLaunchKernelGGL(k_tmvp_cells, dim3((n_units + 255) / 256), dim3(256), 0, ctx->stream, d_units, n_units, d_refined, log2_ctu_s, nb_ctb_w, d_out);
    OV_LAUNCH_CHECK(ctx, "k_tmvp_cells");
    return OVHIP_OK;
}
