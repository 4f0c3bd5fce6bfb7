// kernels_mc.hip -- K5/K6/K11 on gfx950: luma 8-tap / chroma 4-tap separable motion-compensated
// interpolation, uni-prediction, bi-prediction average, BCW weighted bi-prediction, LMCS forward
// reshaping of the luma prediction.
//
// One wavefront (= one 64-thread workgroup) per MC unit (<= 16x16 luma + its 4:2:0 chroma).
// For each reference list and plane the (w+T-1) x (h+T-1) reference window is staged in LDS
// with clamped coordinates (the device equivalent of emulate_block_border(),
// libovvc/rcn_inter.c:148-225 -- reference pictures are tight planes, ovframe.h:84-124), rows
// laid out so consecutive lanes read consecutive samples of a reference row; the horizontal
// pass runs LDS->LDS into a 14-bit int16 tile, the vertical pass LDS->registers, and the
// uni/bi/BCW combine + clip + (LMCS) + store happens in registers.  The four reference variants
// (copy / h / v / hv) are ONE separable filter whose integer-phase row is the identity tap
// (vvc_mc_taps.h); int16 x int8 -> int32 on the VALU, no MFMA.
//
// Replaces mc_l/mc_c.{unidir,bidir0,bidir1,bidir_w}[hv][] and their drivers rcn_mcp_l/_c,
// rcn_motion_compensation_b_l/_c (libovvc/rcn_mc.c:382-1610; rcn_inter.c:520-602, :1391-1554,
// :1822-1904) and lmcs_reshape_forward (rcn_lmcs.c:275-295).
#include "ovvc_common.hip.h"
#define OVT_ATTR __device__
#include "vvc_mc_taps.h"

namespace {

#define MC_MAX_REFS 16
struct RefTable { ovhip_pic p[MC_MAX_REFS]; };

#define WIN_STRIDE 24   /* (16 + 7) rounded up */

template <int NT>
__device__ __forceinline__ void predict14(const uint16_t *__restrict__ ref, int rstride, int rw, int rh,
                                           int px, int py, int log2w, int h, const int8_t *fh, const int8_t *fv,
                                           uint16_t *s_win, int16_t *s_h, int lane, int P[4])
{
    constexpr int before = NT == 8 ? 3 : 1;
    const int w = 1 << log2w;
    const int ww = w + NT - 1, wh = h + NT - 1;

    int th[NT], tv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { th[t] = fh[t]; tv[t] = fv[t]; }

    for (int i = lane; i < ww * wh; i += 64) {
        const int r = i / ww, c = i - r * ww;
        const int sy = ov_clip3(py + r - before, 0, rh - 1);
        const int sx = ov_clip3(px + c - before, 0, rw - 1);
        s_win[r * WIN_STRIDE + c] = ref[sy * rstride + sx];
    }
    __syncthreads();
    // horizontal: t = F_h(src) >> (BITDEPTH - 8)
    for (int i = lane; i < (wh << log2w); i += 64) {
        const int r = i >> log2w, x = i & (w - 1);
        const uint16_t *s = s_win + r * WIN_STRIDE + x;
        int acc = 0;
#pragma unroll
        for (int t = 0; t < NT; ++t) acc += th[t] * (int)s[t];
        s_h[r * 16 + x] = (int16_t)(acc >> (OV_BD - 8));
    }
    __syncthreads();
    // vertical: P = F_v(t) >> 6   (14-bit intermediate)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = lane + 64 * q;
        if (i < (h << log2w)) {
            const int y = i >> log2w, x = i & (w - 1);
            int acc = 0;
#pragma unroll
            for (int t = 0; t < NT; ++t) acc += tv[t] * (int)s_h[(y + t) * 16 + x];
            P[q] = acc >> 6;
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(64) void k_mc(ovhip_pic dst, RefTable refs, const ovhip_mc_unit *__restrict__ units,
                                            uint32_t n_units, const uint16_t *__restrict__ lmcs_fwd)
{
    __shared__ uint16_t s_win[(16 + 7) * WIN_STRIDE];
    __shared__ int16_t s_h[(16 + 7) * 16];

    const uint32_t bid = blockIdx.x;
    if (bid >= n_units) return;
    const int lane = threadIdx.x;
    const ovhip_mc_unit u = units[bid];

    const int first_plane = (u.flags & OVHIP_MC_NO_LUMA) ? 1 : 0;
    const int last_plane = (u.flags & OVHIP_MC_NO_CHROMA) ? 0 : 2;

    for (int plane = first_plane; plane <= last_plane; ++plane) {
        const int c = plane != 0;
        const int w = u.w >> c, h = u.h >> c;
        const int log2w = 31 - __clz(w);
        const int x0 = u.x >> c, y0 = u.y >> c;
        int P[2][4];

#pragma unroll
        for (int l = 0; l < 2; ++l) {
            if (!(u.dir & (1 << l))) continue;
            const ovhip_pic &rp = refs.p[l ? u.ref1 : u.ref0];
            const int mvx = l ? u.mv1x : u.mv0x, mvy = l ? u.mv1y : u.mv0y;
            if (!c) {
                int fx = mvx & 15, fy = mvy & 15;
                const int8_t *fh, *fv;
                if (u.flags & OVHIP_MC_FILT_4x4) { fh = ovt_mc_luma4[fx]; fv = ovt_mc_luma4[fy]; }
                else {
                    if (u.flags & OVHIP_MC_HPEL_FILT) { if (fx == 8) fx = 16; if (fy == 8) fy = 16; }
                    fh = ovt_mc_luma[fx]; fv = ovt_mc_luma[fy];
                }
                predict14<8>(rp.y, rp.stride_y, rp.w, rp.h, x0 + (mvx >> 4), y0 + (mvy >> 4), log2w, h,
                             fh, fv, s_win, s_h, lane, P[l]);
            } else {
                const uint16_t *r = plane == 1 ? rp.cb : rp.cr;
                predict14<4>(r, rp.stride_c, rp.w >> 1, rp.h >> 1, x0 + (mvx >> 5), y0 + (mvy >> 5), log2w, h,
                             ovt_mc_chroma[mvx & 31], ovt_mc_chroma[mvy & 31], s_win, s_h, lane, P[l]);
            }
        }

        int dstride;
        uint16_t *d = ov_plane(dst, plane, dstride) + y0 * dstride + x0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = lane + 64 * q;
            if (i < (h << log2w)) {
                const int y = i >> log2w, x = i & (w - 1);
                int v;
                if (u.dir != 3)                    v = ov_clip_bd(((u.dir == 1 ? P[0][q] : P[1][q]) + 8) >> 4);
                else if (u.w0 == 4 && u.w1 == 4)   v = ov_clip_bd((P[0][q] + P[1][q] + 16) >> 5);
                else                               v = ov_clip_bd((P[1][q] * u.w1 + P[0][q] * u.w0 + 64) >> 7);
                if (!c && (u.flags & OVHIP_MC_LMCS) && lmcs_fwd) v = lmcs_fwd[v];
                d[y * dstride + x] = (uint16_t)v;
            }
        }
    }
}

} // namespace

extern "C" int ovhip_mc_launch(ovhip_ctx *ctx, const ovhip_pic *dst, const ovhip_pic *refs, uint32_t n_refs,
                               const ovhip_mc_unit *d_units, uint32_t n_units, const uint16_t *d_lmcs_fwd_lut)
{
    if (!ctx || !dst) return OVHIP_EINVAL;
    if (!n_units) return OVHIP_OK;
    if (!refs || !n_refs || n_refs > MC_MAX_REFS || !d_units)
        return ov_fail(ctx, OVHIP_EINVAL, "ovhip_mc_launch: bad reference table / units", hipSuccess);
    RefTable t;
    memset(&t, 0, sizeof(t));
    for (uint32_t i = 0; i < n_refs; ++i) t.p[i] = refs[i];
    for (uint32_t i = n_refs; i < MC_MAX_REFS; ++i) t.p[i] = refs[0];
    hipLaunchKernelGGL(k_mc, dim3(n_units), dim3(64), 0, ctx->stream, *dst, t, d_units, n_units, d_lmcs_fwd_lut);
    OV_LAUNCH_CHECK(ctx, "k_mc");
    return OVHIP_OK;
}
