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

// Occupancy hint (waves per SIMD the register allocator must leave room for; 0 = compiler default).
// -DOV_WPE_MC=n overrides it for sweeps.
#ifndef OV_WPE_MC
#define OV_WPE_MC 8       /* measured (tools/sweep_occupancy.sh): k_mc2 48.5 -> 46.5 us */
#endif
#if OV_WPE_MC > 0
#define OV_OCC_MC __attribute__((amdgpu_waves_per_eu(OV_WPE_MC)))
#else
#define OV_OCC_MC
#endif

namespace {

// geometric partitioning: w = clip3(0, 8, (K + A*x + B*y) >> 3), put_weighted_gpm_bi_pixels (rcn_mc.c:1630-1655)
__device__ __forceinline__ int mc_gpm(uint32_t aux, int x, int y, int p0, int p1)
{
    const int k = (int16_t)(aux & 0xffff), a = (int8_t)((aux >> 16) & 0xff), b = (int8_t)(aux >> 24);
    const int wgt = ov_clip3((k + __mul24(a, x) + __mul24(b, y)) >> 3, 0, 8);
    return ov_clip_bd((__mul24(p1, 8 - wgt) + __mul24(p0, wgt) + 64) >> 7);
}

// uni (p + 8) >> 4, bi average (p0 + p1 + 16) >> 5, BCW (p1 w1 + p0 w0 + 64) >> 7 (put_vvc_uni/bi(_w)_*): one
// weighted form whose constants are chosen ONCE per unit -- per-sample three-way branches on wave-uniform unit
// fields cost scalar issue slots, which is what bounds this kernel
struct Combine { int a0, a1, rnd, sh; };
__device__ __forceinline__ Combine mc_combine_of(const ovhip_mc_unit &u)
{
    Combine c;
    if (u.dir != 3)                 { c.a0 = u.dir == 1; c.a1 = u.dir != 1; c.rnd = 8; c.sh = 4; }
    else if (u.w0 == 4 && u.w1 == 4) { c.a0 = 1; c.a1 = 1; c.rnd = 16; c.sh = 5; }
    else                            { c.a0 = u.w0; c.a1 = u.w1; c.rnd = 64; c.sh = 7; }
    return c;
}
// (24-bit multiplies: the 14-bit intermediates and the weights fit, and v_mad_i32_i24 is full rate where v_mul_lo_u32 is not)
__device__ __forceinline__ int mc_combine(const Combine &c, int p0, int p1) { return ov_clip_bd((__mul24(p1, c.a1) + __mul24(p0, c.a0) + c.rnd) >> c.sh); }

// =====================================================================================================
// k_mc2 (the first version, k_mc, ran one pass per list and per plane with 4 rows per lane: SQ counters showed it
// bound by VALU issue at ~57 % of the chip with most passes at 10-50 % lane occupancy for 8x8 / 16x8 units; this
// layout keeps the wave's 64 lanes busy for every unit shape, 82 -> 67 us at 4K):
//   * horizontal pass: ONE loop over the tasks of both lists (luma), one over both lists x both chroma planes
//   * vertical pass: lane = (column, group of NOUT rows) with NOUT = max(1, w*h/64) so that 64 lanes cover
//     the whole block once; each lane runs both lists for its samples and combines in registers; chroma:
//     both planes in the same pass
// =====================================================================================================
// fused CIIP blend: (intra * wt + inter * (4 - wt) + 2) >> 2, put_weighted_ciip_pixels (rcn_mc.c:1611-1628)
__device__ __forceinline__ int ciip_blend(int inter, int intra, int wt) { return ov_clip_bd((__mul24(intra, wt) + __mul24(inter, 4 - wt) + 2) >> 2); }

#define HLS 26   /* k_mc2's transposed luma H tile: odd dword stride (conflict-free columns), rows h + 7 <= 23 */
template <int NOUT>
__device__ __forceinline__ void luma_finish(const ovhip_mc_unit &u, const ovhip_pic &dst, const int16_t *s_hl, const int tv[2][4],
                                            int lane, int log2w, const uint16_t *__restrict__ lmcs_fwd, const ovhip_pic &intra)
{
    const int w = 1 << log2w;
    const int x = lane & (w - 1), yg = lane >> log2w, y0 = yg * NOUT;
    if (y0 >= u.h) return;
    int P[2][NOUT];
#pragma unroll
    for (int l = 0; l < 2; ++l) {
#pragma unroll
        for (int o = 0; o < NOUT; ++o) P[l][o] = 0;
        if (u.dir & (1 << l)) v_outputs<8, NOUT>(s_hl + l * 16 * HLS + x * HLS, y0, tv[l], P[l]);
    }
    uint16_t *d = dst.y + ov_rowoff(u.y + y0, dst.stride_y) + u.x + x;
    // every wave-uniform decision once, not once per sample
    int v[NOUT];
    if (u.flags & OVHIP_MC_GPM) {
#pragma unroll
        for (int j = 0; j < NOUT; ++j) v[j] = mc_gpm(u.aux, x, y0 + j, P[0][j], P[1][j]);
    } else {
        const Combine cmb = mc_combine_of(u);
#pragma unroll
        for (int j = 0; j < NOUT; ++j) v[j] = mc_combine(cmb, P[0][j], P[1][j]);
    }
    if ((u.flags & OVHIP_MC_LMCS) && lmcs_fwd) {
#pragma unroll
        for (int j = 0; j < NOUT; ++j) v[j] = lmcs_fwd[v[j]];
    }
    if (!(u.flags & OVHIP_MC_GPM) && u.aux) {
        const uint16_t *ip = intra.y + ov_rowoff(u.y + y0, intra.stride_y) + u.x + x;
#pragma unroll
        for (int j = 0; j < NOUT; ++j) v[j] = ciip_blend(v[j], ip[j * intra.stride_y], u.aux & 7);
    }
#pragma unroll
    for (int j = 0; j < NOUT; ++j) d[j * dst.stride_y] = (uint16_t)v[j];
}

template <int NOUT>
__device__ __forceinline__ void chroma_finish(const ovhip_mc_unit &u, const ovhip_pic &dst, const int16_t *s_hc, const int tv[2][2],
                                              int lane, int log2wc, int hc, const ovhip_pic &intra)
{
    const int wc = 1 << log2wc;
    const int per_plane = (wc * hc) / NOUT;                   // lanes per plane (power of two, <= 32)
    const int plane = lane >= per_plane, ll = lane - plane * per_plane;
    if (lane >= 2 * per_plane) return;
    const int x = ll & (wc - 1), y0 = (ll >> log2wc) * NOUT;
    int P[2][NOUT];
#pragma unroll
    for (int l = 0; l < 2; ++l) {
#pragma unroll
        for (int o = 0; o < NOUT; ++o) P[l][o] = 0;
        if (u.dir & (1 << l)) v_outputs<4, NOUT>(s_hc + (plane * 2 + l) * 8 * CHT_STRIDE + x * CHT_STRIDE, y0, tv[l], P[l]);
    }
    uint16_t *d = (plane ? dst.cr : dst.cb) + ov_rowoff((u.y >> 1) + y0, dst.stride_c) + (u.x >> 1) + x;
    const bool ciip = !(u.flags & OVHIP_MC_GPM) && u.aux && !(u.aux & 0x100);
    const uint16_t *ip = ciip ? (plane ? intra.cr : intra.cb) + ov_rowoff((u.y >> 1) + y0, intra.stride_c) + (u.x >> 1) + x : nullptr;
    int v[NOUT];
    if (u.flags & OVHIP_MC_GPM) {
#pragma unroll
        for (int j = 0; j < NOUT; ++j) v[j] = mc_gpm(u.aux, 2 * x, 2 * (y0 + j), P[0][j], P[1][j]);
    } else {
        const Combine cmb = mc_combine_of(u);
#pragma unroll
        for (int j = 0; j < NOUT; ++j) v[j] = mc_combine(cmb, P[0][j], P[1][j]);
    }
    if (ciip) {
#pragma unroll
        for (int j = 0; j < NOUT; ++j) v[j] = ciip_blend(v[j], ip[j * intra.stride_c], u.aux & 7);
    }
#pragma unroll
    for (int j = 0; j < NOUT; ++j) d[j * dst.stride_c] = (uint16_t)v[j];
}

#ifdef OV_MC_PHASES
// Debug build only (-DOV_MC_PHASES): per-phase shader-clock totals of k_mc2, summed over waves.
#define OV_MC_PHASE_UNITS 65536
__device__ unsigned int g_mc_phase[OV_MC_PHASE_UNITS * 8];
#define OV_PHASE(i) do { __builtin_amdgcn_s_waitcnt(0); const unsigned long long t_ = __builtin_readcyclecounter(); \
                         ph[i] = t_ - tprev; tprev = t_; } while (0)
#else
#define OV_PHASE(i) do { } while (0)
#endif

__global__ __launch_bounds__(64) OV_OCC_MC void k_mc2(ovhip_pic dst, RefTable refs, const ovhip_mc_unit *__restrict__ units,
                                             uint32_t n_units, const uint16_t *__restrict__ lmcs_fwd, int xcd, ovhip_pic intra)
{
    // One LDS block of 5024 B -- under the 5120 B step at which a CU holds 32 of these single-wave workgroups (measured:
    // 5632..6400 B -> 24, 6608 B -> 21):  [ luma windows, list 0 / 1 | chroma windows, later the luma H tiles | chroma H tiles ].
    // The chroma H pass runs first so that the luma H tiles can take over the chroma windows.  Whole-dword over-reads
    // past a region land in the next one (or the slack at the end) and are never used.
    constexpr int HL_STRIDE = HLS;
    constexpr int R2 = 2 * 16 * HL_STRIDE > 4 * CHR_WIN ? 2 * 16 * HL_STRIDE : 4 * CHR_WIN;
    __shared__ __attribute__((aligned(16))) uint16_t s_all[2 * LUMA_WIN + R2 + 4 * 8 * CHT_STRIDE + 8];
    uint16_t *const s_wl = s_all;                                                      // [list]
    uint16_t *const s_wc = s_all + 2 * LUMA_WIN;                                       // [plane * 2 + list]
    int16_t  *const s_hl = reinterpret_cast<int16_t *>(s_all + 2 * LUMA_WIN);          // [list], aliases s_wc
    int16_t  *const s_hc = reinterpret_cast<int16_t *>(s_all + 2 * LUMA_WIN + R2);     // [plane * 2 + list]
    static_assert(sizeof(s_all) <= 5120, "k_mc2 LDS block must stay under the 32-workgroups-per-CU step");

    // one unit per workgroup (no grid-stride loop: the loop-carried scalars cost SGPRs this kernel does not have)
    const int lane = threadIdx.x;
    {
    const uint32_t wg = blockIdx.x;
    if (wg >= n_units) return;
    const uint32_t bid = xcd ? ov_xcd_slot(wg, n_units) : wg;
#ifdef OV_MC_PHASES
    unsigned long long ph[8] = {}, tprev = __builtin_readcyclecounter();
#endif
    const ovhip_mc_unit u = units[bid];
    OV_PHASE(0);

    const bool do_l = !(u.flags & OVHIP_MC_NO_LUMA), do_c = !(u.flags & OVHIP_MC_NO_CHROMA);
    const int w = u.w, h = u.h, wc = w >> 1, hc = h >> 1;
    const int log2w = 31 - __clz(w), log2wc = log2w - 1;
    const int nl = u.dir == 3 ? 2 : 1, l0 = u.dir == 2 ? 1 : 0;      // lists present: l0 .. l0 + nl - 1

    // ---- windows: issue all loads, then park.  Every reference picture has dst's geometry (checked at launch), so
    // only the three plane pointers of each list come from the table -- fetched for both lists before first use. ----
    const int ri0 = (u.dir & 1) ? u.ref0 : u.ref1, ri1 = (u.dir & 2) ? u.ref1 : u.ref0;
    const uint16_t *const ry[2]  = { refs.p[ri0].y,  refs.p[ri1].y };
    const uint16_t *const rcb[2] = { refs.p[ri0].cb, refs.p[ri1].cb };
    const uint16_t *const rcr[2] = { refs.p[ri0].cr, refs.p[ri1].cr };
    int lx[2], ly[2], cx[2], cy[2];
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        const int mvx = l ? u.mv1x : u.mv0x, mvy = l ? u.mv1y : u.mv0y;
        lx[l] = u.x + (mvx >> 4) - 3;        ly[l] = u.y + (mvy >> 4) - 3;
        cx[l] = (u.x >> 1) + (mvx >> 5) - 1; cy[l] = (u.y >> 1) + (mvy >> 5) - 1;
    }
    // Horizontal taps: every lane works for ONE window in the H passes (see there), so it fetches that list's filter itself
    // -- a broadcast vector load issued here, ahead of the window loads, instead of both lists' taps in SGPRs and a select
    // per tap (the kernel is short of SGPRs; vertical taps, which every lane needs for both lists, stay scalar).
    const uint32_t (*const ltab)[4] = (u.flags & OVHIP_MC_FILT_4x4) ? g_taps.luma4 : g_taps.luma;
    const bool hpel = !(u.flags & OVHIP_MC_FILT_4x4) && (u.flags & OVHIP_MC_HPEL_FILT);
    const int hl_l = l0 + (nl == 2 ? lane >> 5 : 0);                          // list of this lane's luma window
    const int hc_q = lane >> (nl == 2 ? 4 : 5);                               // chroma window: plane * 2 + list (2 lists) / plane
    const int hc_l = l0 + (nl == 2 ? (hc_q & 1) : 0);
    int fxl = (hl_l ? u.mv1x : u.mv0x) & 15;
    if (hpel && fxl == 8) fxl = 16;
    const int fxc = (hc_l ? u.mv1x : u.mv0x) & 31;
    const uint4 hl_tp = *reinterpret_cast<const uint4 *>(ltab[fxl]);
    const uint2 hc_tp = *reinterpret_cast<const uint2 *>(g_taps.chroma[fxc]);
    uint16_t *const lwin[2] = { s_wl, s_wl + LUMA_WIN };
    uint16_t *const cwin[4] = { s_wc, s_wc + CHR_WIN, s_wc + 2 * CHR_WIN, s_wc + 3 * CHR_WIN };
    int offl[2], offc1[2];
    stage_unit_windows(dst, ry, rcb, rcr, lx, ly, cx, cy, w, h, u.dir, do_l, do_c, lane, lwin, WIN_STRIDE, cwin, CWIN_STRIDE, offl, offc1);
    const int offc[2][2] = { { offc1[0], offc1[1] }, { offc1[0], offc1[1] } };
    // ---- filter taps of both lists (wave-uniform) ----
    __syncthreads();
    OV_PHASE(2);

    // ---- horizontal passes: one task = one window row x 4 outputs.  Chroma first (see the LDS layout).
    // The lanes are dealt to the windows in equal groups (luma: 64 or 2 x 32 lanes, chroma: 2 x 32 or 4 x 16), so that
    // everything a task needs except its row -- list, taps, alignment, tile -- is fixed per lane and leaves the loop
    // (a loop over one merged task list spent a quarter of this kernel's vector instructions on that bookkeeping); the
    // iteration counts are the same as for the merged list in every common unit shape. ----
    if (do_c) {
        const int log2seg = log2wc > 2 ? log2wc - 2 : 0;
        const int l2per = nl == 2 ? 4 : 5;                                // lanes per window: 16 (4 windows) or 32 (2)
        const int qi = hc_q, tl = lane & ((1 << l2per) - 1);
        const int plane = nl == 2 ? qi >> 1 : qi, l = hc_l;
        const int tp[2] = { (int)hc_tp.x, (int)hc_tp.y };
        const int off = plane ? (l ? offc[1][1] : offc[1][0]) : (l ? offc[0][1] : offc[0][0]);
        const bool ident = fxc == 0;
        const uint16_t *src = s_wc + (plane * 2 + l) * CHR_WIN;
        int16_t *ht = s_hc + (plane * 2 + l) * 8 * CHT_STRIDE;
        const int x0 = (tl & ((1 << log2seg) - 1)) << 2, rstep = (1 << l2per) >> log2seg;
        for (int r = tl >> log2seg; r < hc + 3; r += rstep) h_task<4>(src + r * CWIN_STRIDE, off, x0, tp, ident, ht, CHT_STRIDE, r);
    }
    __syncthreads();          // the chroma windows are dead: s_hl takes their place
    if (do_l) {
        const int log2seg = log2w > 2 ? log2w - 2 : 0;
        const int l2per = nl == 2 ? 5 : 6;                                // lanes per window: 32 (2 lists) or 64
        const int tl = lane & ((1 << l2per) - 1), l = hl_l;
        const int tp[4] = { (int)hl_tp.x, (int)hl_tp.y, (int)hl_tp.z, (int)hl_tp.w };
        const int off = l ? offl[1] : offl[0];
        const bool ident = fxl == 0;
        const uint16_t *src = s_wl + l * LUMA_WIN;
        int16_t *ht = s_hl + l * 16 * HL_STRIDE;
        const int x0 = (tl & ((1 << log2seg) - 1)) << 2, rstep = (1 << l2per) >> log2seg;
        for (int r = tl >> log2seg; r < h + 7; r += rstep) h_task<8>(src + r * WIN_STRIDE, off, x0, tp, ident, ht, HL_STRIDE, r);
    }
    __syncthreads();
    OV_PHASE(3);

    // ---- vertical passes + combine + store: every lane finishes NOUT samples of one column ----
    int tvl[2][4], tvc[2][2];
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        const int mvy = l ? u.mv1y : u.mv0y;
        int fy = mvy & 15;
        if (hpel && fy == 8) fy = 16;
        load_taps<4>(ltab[fy], tvl[l]);
        load_taps<2>(g_taps.chroma[mvy & 31], tvc[l]);
    }
    if (do_l) {
        const int npix = w * h;
        if (npix >= 256)      luma_finish<4>(u, dst, s_hl, tvl, lane, log2w, lmcs_fwd, intra);
        else if (npix >= 128) luma_finish<2>(u, dst, s_hl, tvl, lane, log2w, lmcs_fwd, intra);
        else                  luma_finish<1>(u, dst, s_hl, tvl, lane, log2w, lmcs_fwd, intra);
    }
    if (do_c) {
        if (wc * hc >= 64) chroma_finish<2>(u, dst, s_hc, tvc, lane, log2wc, hc, intra);
        else               chroma_finish<1>(u, dst, s_hc, tvc, lane, log2wc, hc, intra);
    }
    __syncthreads();          // LDS tiles are reused by the next unit
#ifdef OV_MC_PHASES
    { const unsigned long long t_ = __builtin_readcyclecounter(); ph[4] = t_ - tprev; tprev = t_; }   // V pass, stores issued
    OV_PHASE(5);                                                                                      // stores drained
    if (lane == 0 && bid < OV_MC_PHASE_UNITS) {
#pragma unroll
        for (int i = 0; i < 6; ++i) g_mc_phase[bid * 8 + i] = (unsigned int)ph[i];
        g_mc_phase[bid * 8 + 7] = 1;
    }
#endif
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

#ifdef OV_MC_PHASES
extern "C" int ovhip_debug_mc_phases(unsigned int *out /* [65536][8] */)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mc_phase), sizeof(unsigned int) * OV_MC_PHASE_UNITS * 8) == hipSuccess ? OVHIP_OK : OVHIP_ELAUNCH;
}
#endif

extern "C" int ovhip_ciip_launch(ovhip_ctx *ctx, const ovhip_pic *dst, const ovhip_pic *intra,
                                 const ovhip_ciip_unit *d_units, uint32_t n_units)
{
    if (!ctx || !dst || !intra) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    if (!n_units) return OVHIP_OK;
    if (!d_units) return ov_fail(ctx, OVHIP_EINVAL, "ovhip_ciip_launch: null units", hipSuccess);
    hipLaunchKernelGGL(k_ciip, dim3(n_units), dim3(256), 0, ctx->stream, *dst, *intra, d_units, n_units);
    OV_LAUNCH_CHECK(ctx, "k_ciip");
    return OVHIP_OK;
}

extern "C" int ovhip_mc_launch(ovhip_ctx *ctx, const ovhip_pic *dst, const ovhip_pic *refs, uint32_t n_refs,
                               const ovhip_mc_unit *d_units, uint32_t n_units, const uint16_t *d_lmcs_fwd_lut,
                               const ovhip_pic *intra)
{
    if (!ctx || !dst) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    if (!n_units) return OVHIP_OK;
    if (!refs || !n_refs || n_refs > MC_MAX_REFS || !d_units)
        return ov_fail(ctx, OVHIP_EINVAL, "ovhip_mc_launch: bad reference table / units", hipSuccess);
    RefTable t;
    memset(&t, 0, sizeof(t));
    for (uint32_t i = 0; i < n_refs; ++i) {
        // k_mc2 takes the window geometry from dst: references of another size are RPR, outside this path
        if (refs[i].w != dst->w || refs[i].h != dst->h || refs[i].stride_y != dst->stride_y || refs[i].stride_c != dst->stride_c)
            return ov_fail(ctx, OVHIP_EUNSUP, "ovhip_mc_launch: reference picture geometry differs from dst (RPR)", hipSuccess);
        t.p[i] = refs[i];
    }
    for (uint32_t i = n_refs; i < MC_MAX_REFS; ++i) t.p[i] = refs[0];
    static int cfg_xcd = -1;
#ifdef OVHIP_TUNING
    if (cfg_xcd < 0) { const char *x = getenv("OVHIP_MC_XCD"); cfg_xcd = x ? atoi(x) : 1; }   // experiment knob: XCD-aware unit order
#else
    cfg_xcd = 1;
#endif
    // one single-wave workgroup per unit (measured faster than a resident grid-stride grid).  Units with a fused CIIP
    // blend read `intra`; without such units the argument is never dereferenced.
    hipLaunchKernelGGL(k_mc2, dim3(n_units), dim3(64), 0, ctx->stream, *dst, t, d_units, n_units, d_lmcs_fwd_lut, cfg_xcd,
                       intra ? *intra : *dst);
    OV_LAUNCH_CHECK(ctx, "k_mc2");
    return OVHIP_OK;
}
