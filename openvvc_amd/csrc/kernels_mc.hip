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
#include "mc_common.hip.h"
#include <stdlib.h>

namespace {

// geometric partitioning: w = clip3(0, 8, (K + A*x + B*y) >> 3), put_weighted_gpm_bi_pixels (rcn_mc.c:1630-1655)
__device__ __forceinline__ int mc_gpm(uint32_t aux, int x, int y, int p0, int p1)
{
    const int k = (int16_t)(aux & 0xffff), a = (int8_t)((aux >> 16) & 0xff), b = (int8_t)(aux >> 24);
    const int wgt = ov_clip3((k + a * x + b * y) >> 3, 0, 8);
    return ov_clip_bd((p1 * (8 - wgt) + p0 * wgt + 64) >> 7);
}

__device__ __forceinline__ int mc_combine(const ovhip_mc_unit &u, int p0, int p1)
{
    if (u.dir != 3)                  return ov_clip_bd(((u.dir == 1 ? p0 : p1) + 8) >> 4);
    if (u.w0 == 4 && u.w1 == 4)      return ov_clip_bd((p0 + p1 + 16) >> 5);
    return ov_clip_bd((p1 * u.w1 + p0 * u.w0 + 64) >> 7);
}

__global__ __launch_bounds__(64) void k_mc(ovhip_pic dst, RefTable refs, const ovhip_mc_unit *__restrict__ units,
                                            uint32_t n_units, const uint16_t *__restrict__ lmcs_fwd, int ablate, int xcd)
{
    __shared__ __attribute__((aligned(16))) uint16_t s_wl[2][LUMA_WIN];          // luma windows, list 0 / 1
    __shared__ __attribute__((aligned(16))) uint16_t s_wc[2][2][CHR_WIN];        // chroma windows [Cb/Cr][list]
    __shared__ __attribute__((aligned(16))) int16_t  s_hl[2][16 * HT_STRIDE];    // transposed H-pass tiles
    __shared__ __attribute__((aligned(16))) int16_t  s_hc[2][2][8 * CHT_STRIDE];

    const int lane = threadIdx.x;
    // grid-stride over units: the chip launches ~400 workgroups/us, so one workgroup per unit would be
    // dispatch-bound; a resident grid of single-wave workgroups walks the unit list instead
    for (uint32_t wg = blockIdx.x; wg < n_units; wg += gridDim.x) {
    // XCD-aware order: workgroup i runs on XCD i % 8, so XCD k walks the k-th contiguous eighth of the
    // unit list (= a compact area of the picture) and neighbouring reference windows meet in ITS L2
    const uint32_t bid = xcd ? ov_xcd_slot(wg, n_units) : wg;
    const ovhip_mc_unit u = units[bid];

    const bool do_l = !(u.flags & OVHIP_MC_NO_LUMA), do_c = !(u.flags & OVHIP_MC_NO_CHROMA);
    const int w = u.w, h = u.h, wc = w >> 1, hc = h >> 1;
    const int log2w = 31 - __clz(w), log2wc = log2w - 1;

    // ---- issue all loads, then park ----
    LumaStage sl[2];
    ChromaStage sc[2][2];
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        if (!(u.dir & (1 << l))) continue;
        const ovhip_pic &rp = refs.p[l ? u.ref1 : u.ref0];
        const int mvx = l ? u.mv1x : u.mv0x, mvy = l ? u.mv1y : u.mv0y;
        if (ablate & 1) continue;
        if (do_l) sl[l].issue(rp.y, rp.stride_y, rp.w, rp.h, u.x + (mvx >> 4) - 3, u.y + (mvy >> 4) - 3, w + 7, h + 7, lane, s_wl[l], WIN_STRIDE);
        if (do_c) {
            const int px = (u.x >> 1) + (mvx >> 5) - 1, py = (u.y >> 1) + (mvy >> 5) - 1;
            sc[0][l].issue(rp.cb, rp.stride_c, rp.w >> 1, rp.h >> 1, px, py, wc + 3, hc + 3, lane, s_wc[0][l], CWIN_STRIDE);
            sc[1][l].issue(rp.cr, rp.stride_c, rp.w >> 1, rp.h >> 1, px, py, wc + 3, hc + 3, lane, s_wc[1][l], CWIN_STRIDE);
        }
    }
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        if (!(u.dir & (1 << l)) || (ablate & 1)) continue;
        if (do_l) sl[l].park(s_wl[l], WIN_STRIDE, w + 7, h + 7, lane);
        if (do_c) {
            sc[0][l].park(s_wc[0][l], CWIN_STRIDE, wc + 3, hc + 3, lane);
            sc[1][l].park(s_wc[1][l], CWIN_STRIDE, wc + 3, hc + 3, lane);
        }
    }
    __syncthreads();

    // ---- horizontal passes ----
    const int8_t *fvl[2], *fvc[2];
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        if (!(u.dir & (1 << l))) continue;
        const int mvx = l ? u.mv1x : u.mv0x, mvy = l ? u.mv1y : u.mv0y;
        int fx = mvx & 15, fy = mvy & 15;
        const int8_t *fh;
        if (u.flags & OVHIP_MC_FILT_4x4) { fh = ovt_mc_luma4[fx]; fvl[l] = ovt_mc_luma4[fy]; }
        else {
            if (u.flags & OVHIP_MC_HPEL_FILT) { if (fx == 8) fx = 16; if (fy == 8) fy = 16; }
            fh = ovt_mc_luma[fx]; fvl[l] = ovt_mc_luma[fy];
        }
        fvc[l] = ovt_mc_chroma[mvy & 31];
        if (ablate & 2) continue;
        if (do_l) h_pass<8>(s_wl[l], WIN_STRIDE, sl[l].off, s_hl[l], HT_STRIDE, log2w, h + 7, fh, lane);
        if (do_c) {
            h_pass<4>(s_wc[0][l], CWIN_STRIDE, sc[0][l].off, s_hc[0][l], CHT_STRIDE, log2wc, hc + 3, ovt_mc_chroma[mvx & 31], lane);
            h_pass<4>(s_wc[1][l], CWIN_STRIDE, sc[1][l].off, s_hc[1][l], CHT_STRIDE, log2wc, hc + 3, ovt_mc_chroma[mvx & 31], lane);
        }
    }
    __syncthreads();

    // ---- vertical passes + combine + store ----
    if (ablate & 4) continue;
    if (do_l) {
        int P[2][4];
#pragma unroll
        for (int l = 0; l < 2; ++l) if (u.dir & (1 << l)) v_pass<8>(s_hl[l], HT_STRIDE, log2w, h, fvl[l], lane, P[l]);
        if (lane < (((h + 3) >> 2) << log2w)) {
            const int x = lane & (w - 1), g = lane >> log2w;
            uint16_t *d = dst.y + (u.y + 4 * g) * dst.stride_y + u.x + x;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (4 * g + j < h) {
                    int v = (u.flags & OVHIP_MC_GPM) ? mc_gpm(u.aux, x, 4 * g + j, P[0][j], P[1][j]) : mc_combine(u, P[0][j], P[1][j]);
                    if ((u.flags & OVHIP_MC_LMCS) && lmcs_fwd) v = lmcs_fwd[v];
                    d[j * dst.stride_y] = (uint16_t)v;
                }
            }
        }
    }
    if (do_c) {
#pragma unroll
        for (int comp = 0; comp < 2; ++comp) {
            int P[2][4];
#pragma unroll
            for (int l = 0; l < 2; ++l) if (u.dir & (1 << l)) v_pass<4>(s_hc[comp][l], CHT_STRIDE, log2wc, hc, fvc[l], lane, P[l]);
            if (lane < (((hc + 3) >> 2) << log2wc)) {
                const int x = lane & (wc - 1), g = lane >> log2wc;
                uint16_t *d = (comp ? dst.cr : dst.cb) + ((u.y >> 1) + 4 * g) * dst.stride_c + (u.x >> 1) + x;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (4 * g + j < hc)
                        d[j * dst.stride_c] = (uint16_t)((u.flags & OVHIP_MC_GPM) ? mc_gpm(u.aux, 2 * x, 2 * (4 * g + j), P[0][j], P[1][j])
                                                                                  : mc_combine(u, P[0][j], P[1][j]));
            }
        }
    }
    __syncthreads();          // LDS tiles are reused by the next unit
    }
}

// ---- K10: CIIP blend.  One 256-thread workgroup per CU: dst = (intra * wt + inter * (4 - wt) + 2) >> 2
// (put_weighted_ciip_pixels rcn_mc.c:1611-1628, rcn_ciip_weighted_sum rcn_inter.c:2968-3009). ----
__global__ __launch_bounds__(256) void k_ciip(ovhip_pic dst, ovhip_pic intra, const ovhip_ciip_unit *__restrict__ units, uint32_t n)
{
    for (uint32_t bid = blockIdx.x; bid < n; bid += gridDim.x) {
        const ovhip_ciip_unit u = units[bid];
#pragma unroll
        for (int plane = 0; plane < 3; ++plane) {
            const int c = plane != 0;
            if (c && u.chroma_inter) continue;
            const int lw = u.log2_w - c, w = 1 << lw, npix = w << (u.log2_h - c);
            int ds, is;
            uint16_t *d = ov_plane(dst, plane, ds) + (u.y >> c) * ds + (u.x >> c);
            const uint16_t *s = ov_plane(intra, plane, is) + (u.y >> c) * is + (u.x >> c);
            for (int t = threadIdx.x; t < npix; t += 256) {
                const int x = t & (w - 1), y = t >> lw;
                d[y * ds + x] = (uint16_t)ov_clip_bd((s[y * is + x] * u.wt + d[y * ds + x] * (4 - u.wt) + 2) >> 2);
            }
        }
    }
}

} // namespace

extern "C" int ovhip_ciip_launch(ovhip_ctx *ctx, const ovhip_pic *dst, const ovhip_pic *intra,
                                 const ovhip_ciip_unit *d_units, uint32_t n_units)
{
    if (!ctx || !dst || !intra) return OVHIP_EINVAL;
    if (!n_units) return OVHIP_OK;
    if (!d_units) return ov_fail(ctx, OVHIP_EINVAL, "ovhip_ciip_launch: null units", hipSuccess);
    hipLaunchKernelGGL(k_ciip, dim3(n_units), dim3(256), 0, ctx->stream, *dst, *intra, d_units, n_units);
    OV_LAUNCH_CHECK(ctx, "k_ciip");
    return OVHIP_OK;
}

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
    static int cfg_grid = -1, cfg_ablate = 0, cfg_xcd = 1;
    if (cfg_grid < 0) {                       // developer knobs (profiling experiments only)
        const char *g = getenv("OVHIP_MC_GRID"), *a = getenv("OVHIP_MC_ABLATE"), *x = getenv("OVHIP_MC_XCD");
        cfg_grid = g ? atoi(g) : 0;
        cfg_ablate = a ? atoi(a) : 0;
        cfg_xcd = x ? atoi(x) : 1;
    }
    uint32_t grid = cfg_grid > 0 ? (uint32_t)cfg_grid : n_units;
    if (grid > n_units) grid = n_units;
    hipLaunchKernelGGL(k_mc, dim3(grid), dim3(64), 0, ctx->stream, *dst, t, d_units, n_units, d_lmcs_fwd_lut, cfg_ablate, cfg_grid > 0 ? 0 : cfg_xcd);
    OV_LAUNCH_CHECK(ctx, "k_mc");
    return OVHIP_OK;
}
