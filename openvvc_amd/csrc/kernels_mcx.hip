// kernels_mcx.hip -- K7/K8 on gfx950: "refined" bi-predicted units -- decoder-side motion-vector
// refinement (DMVR) and bi-directional optical flow (BDOF) -- one wavefront per <=16x16 unit.
//
// Replaces rcn_dmvr_mv_refine (libovvc/rcn_inter.c:872-1126: bilinear pre-interpolation
// rcn_mc.c:788-899, 25-point SAD search rcn_inter.c:614-754, parametric sub-pel step :758-833,
// window padding :326-378) and rcn_bdof_mcp_l (rcn_inter.c:1136-1250; gradients, weights and the
// per-4x4 correction rcn_prof_bdof.c:59-103, :152-172, :303-490), plus the chroma of the same CU.
//
// Data flow per unit (everything stays in LDS / registers between the window load and the store):
//   1. both lists' (w+7)x(h+7) luma and (w/2+3)x(h/2+3) chroma windows -> LDS, with a 2-sample apron;
//      DMVR anchors them at clip_mv(initial MV) exactly like derive_dmvr_ref_buf_y/_c
//   2. DMVR: replicate the apron (padd_dmvr), bilinear (w+4)x(h+4) blocks, 25 SADs on every second
//      row (lane = search point x row half), wave arg-min with the reference's tie rule, error-surface
//      step; the refined MVs are lane-uniform and written back for the caller's TMVP field
//   3. 8-tap / 4-tap separable interpolation at the refined position, reading the padded window
//   4. BDOF (unless DMVR's cost test switched it off): 14-bit predictions + integer-sample ring in an
//      18x18 LDS tile, per-sample gradients, 6x6 window sums by 4 lanes per 4x4 block, +-15 weights,
//      corrected average; otherwise the plain average
#include "mc_common.hip.h"
#include <stdlib.h>

// Occupancy hint (waves per SIMD the register allocator must leave room for; 0 = compiler default).
// -DOV_WPE_MCX=n overrides it for sweeps.
#ifndef OV_WPE_MCX
#define OV_WPE_MCX 4
#endif
#if OV_WPE_MCX > 0
#define OV_OCC_MCX __attribute__((amdgpu_waves_per_eu(OV_WPE_MCX)))
#else
#define OV_OCC_MCX
#endif

namespace {

#define XWIN_STRIDE  32    /* 4 (aligned apron slot) + off(<=3) + 23 + 2                                */
#define XCWIN_STRIDE 20    /* 4 + off(<=3) + 11 + 2                                                     */
#define XWIN_ROWS    28    /* 2 + 23 + 2, +1 row of slack for the FIR's whole-dword over-read            */
#define XCWIN_ROWS   16
#define BIL_STRIDE   20
#define R_STRIDE     18

__device__ __forceinline__ void clip_mv_dev(int px, int py, int pic_w, int pic_h, int pw, int ph, int &mvx, int &mvy)
{
    mvx = ov_clip3(mvx, -((pw + 3 + px) << 4), (pic_w + 2 - px) << 4);
    mvy = ov_clip3(mvy, -((ph + 3 + py) << 4), (pic_h + 2 - py) << 4);
}

// replicate a (ww x wh) window held at `win` (row 0 / col 0 = first window sample) 2 samples outwards
__device__ __forceinline__ void pad_sides(uint16_t *win, int stride, int ww, int wh, int lane)
{
    if (lane < wh) {
        uint16_t *r = win + lane * stride;
        const uint16_t a = r[0], b = r[ww - 1];
        r[-1] = a; r[-2] = a; r[ww] = b; r[ww + 1] = b;
    }
}
__device__ __forceinline__ void pad_rows(uint16_t *win, int stride, int ww, int wh, int lane)
{
    if (lane < ww + 4) {
        uint16_t *c = win + lane - 2;
        const uint16_t a = c[0], b = c[(wh - 1) * stride];
        c[-stride] = a; c[-2 * stride] = a; c[wh * stride] = b; c[(wh + 1) * stride] = b;
    }
}

// sum |a - b| over 2 * ND samples of two packed-dword rows (optionally starting one sample into the first dword)
template <int ND>
__device__ __forceinline__ uint32_t sad_row(const int *a, const int *b, bool odd, uint32_t acc)
{
    uint32_t A[ND + 1], B[ND + 1];
#pragma unroll
    for (int k = 0; k < ND + 1; ++k) { A[k] = (uint32_t)a[k]; B[k] = (uint32_t)b[k]; }
    if (odd) {
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            A[k] = __builtin_amdgcn_alignbit(A[k + 1], A[k], 16);
            B[k] = __builtin_amdgcn_alignbit(B[k + 1], B[k], 16);
        }
    }
#pragma unroll
    for (int k = 0; k < ND; ++k) acc = __builtin_amdgcn_sad_u16(A[k], B[k], acc);
    return acc;
}

__device__ __forceinline__ int div_for_maxq7(int num, int den)
{
    int sign = 0, q = 0;
    if (num < 0) { sign = 1; num = -num; }
    den <<= 3;
    if (num >= den) { num -= den; q++; }
    q <<= 1; den >>= 1;
    if (num >= den) { num -= den; q++; }
    q <<= 1;
    if (num >= (den >> 1)) q++;
    return sign ? -q : q;
}

template <int NOUT>
__device__ __forceinline__ void chroma_avg(const ovhip_mc_unit &u, const ovhip_pic &dst, const int16_t *s_hc, const int tv[2][2],
                                           int lane, int log2wc, int hc)
{
    const int wc = 1 << log2wc;
    const int per_plane = (wc * hc) / NOUT;
    const int plane = lane >= per_plane, ll = lane - plane * per_plane;
    if (lane >= 2 * per_plane) return;
    const int x = ll & (wc - 1), y0 = (ll >> log2wc) * NOUT;
    int P[2][NOUT];
#pragma unroll
    for (int l = 0; l < 2; ++l) v_outputs<4, NOUT>(s_hc + (plane * 2 + l) * 8 * CHT_STRIDE + x * CHT_STRIDE, y0, tv[l], P[l]);
    uint16_t *d = (plane ? dst.cr : dst.cb) + ov_rowoff((u.y >> 1) + y0, dst.stride_c) + (u.x >> 1) + x;
#pragma unroll
    for (int j = 0; j < NOUT; ++j) d[j * dst.stride_c] = (uint16_t)ov_clip_bd((P[0][j] + P[1][j] + 16) >> 5);
}

// LDS of one refined unit: luma windows [list], chroma windows [Cb/Cr][list], H-pass tiles
#define MCX_LDS_WL (2 * XWIN_ROWS * XWIN_STRIDE * 2)
#define MCX_LDS_WC (4 * XCWIN_ROWS * XCWIN_STRIDE * 2)
#define MCX_LDS    (MCX_LDS_WL + MCX_LDS_WC + (2 * 16 * HT_STRIDE + 4 * 8 * CHT_STRIDE) * 2)

// units wg0, wg0 + wstride, ... (one single-wave workgroup; `lds` = MCX_LDS bytes, 16-byte aligned)
// SEARCH_ONLY: the decoder-side MV refinement alone (steps 1-2) for the units that carry OVHIP_MC_DMVR, refined vectors
// to mv_out, nothing written to dst -- the eager per-CTU-row pass that keeps the host's TMVP motion field final before
// a row is published (ovhip_job_dmvr_rows).
template <bool SEARCH_ONLY>
__device__ __forceinline__ void mcx_units(const ovhip_pic &dst, const RefTable &refs, const ovhip_mc_unit *__restrict__ units,
                                          uint32_t n_units, const uint16_t *__restrict__ lmcs_fwd, int32_t *__restrict__ mv_out,
                                          uint32_t wg0, uint32_t wstride, char *lds)
{
    uint16_t (*const s_wl)[XWIN_ROWS * XWIN_STRIDE] = reinterpret_cast<uint16_t (*)[XWIN_ROWS * XWIN_STRIDE]>(lds);
    uint16_t (*const s_wc)[2][XCWIN_ROWS * XCWIN_STRIDE] =
        reinterpret_cast<uint16_t (*)[2][XCWIN_ROWS * XCWIN_STRIDE]>(lds + MCX_LDS_WL);        // [Cb/Cr][list]
    // H-pass tiles; the same bytes first hold DMVR's bilinear blocks (dead before the H pass writes) and later BDOF's
    // R tiles (written after the V pass has read the luma tiles: one wave, LDS traffic in program order)
    int16_t *const s_h = reinterpret_cast<int16_t *>(lds + MCX_LDS_WL + MCX_LDS_WC);
    int16_t (*const s_hl)[16 * HT_STRIDE] = reinterpret_cast<int16_t (*)[16 * HT_STRIDE]>(s_h);
    int16_t (*const s_hc)[2][8 * CHT_STRIDE] = reinterpret_cast<int16_t (*)[2][8 * CHT_STRIDE]>(s_h + 2 * 16 * HT_STRIDE);
    int16_t (*const s_x)[24 * BIL_STRIDE] = reinterpret_cast<int16_t (*)[24 * BIL_STRIDE]>(s_h);
    static_assert(2 * 24 * BIL_STRIDE <= 2 * 16 * HT_STRIDE + 4 * 8 * CHT_STRIDE, "bilinear blocks must fit the H tiles");
    // BDOF per-sample terms of derive_bdof_weights, 16x16 each, over the (dead) luma windows:
    uint32_t *const s_ba = reinterpret_cast<uint32_t *>(s_wl[0]);                              // |ax| | |ay| << 16
    uint32_t *const s_bb = s_ba + 256;                                                         // sign(ay) ax | sign(ax) dr << 16
    int16_t  *const s_bc = reinterpret_cast<int16_t *>(s_bb + 256);                            // sign(ay) dr
    static_assert(2 * XWIN_ROWS * XWIN_STRIDE * 2 >= 2 * 1024 + 512, "BDOF terms must fit the luma windows");

    const int lane = threadIdx.x;
    {   // one unit per workgroup (a grid-stride loop's carried scalars cost SGPRs; see k_mc2)
    const uint32_t wg = wg0;
    if (wg >= n_units) return;
    const uint32_t bid = wstride >= n_units ? ov_xcd_slot(wg, n_units) : wg;        // XCD-aware order, see k_mc2
    const ovhip_mc_unit u = units[bid];
    const bool dmvr = u.flags & OVHIP_MC_DMVR;
    if (SEARCH_ONLY && !dmvr) return;
    bool use_bdof = u.flags & OVHIP_MC_BDOF;
    const bool do_l = !(u.flags & OVHIP_MC_NO_LUMA), do_c = !(u.flags & OVHIP_MC_NO_CHROMA);
    const int w = u.w, h = u.h, wc = w >> 1, hc = h >> 1;
    const int log2w = 31 - __clz(w), log2wc = log2w - 1;

    int mv[2][2] = { { u.mv0x, u.mv0y }, { u.mv1x, u.mv1y } };
    int ini[2][2] = { { u.mv0x, u.mv0y }, { u.mv1x, u.mv1y } };

    // ---- 1. windows (reference geometry = dst's, checked at launch) ----
    uint16_t *wl[2], *wcp[2][2];          // first window sample (row 0, col 0) of each staged window
    int offl[2], offc[2];
    {
        const uint16_t *const ry[2]  = { refs.p[u.ref0].y,  refs.p[u.ref1].y };
        const uint16_t *const rcb[2] = { refs.p[u.ref0].cb, refs.p[u.ref1].cb };
        const uint16_t *const rcr[2] = { refs.p[u.ref0].cr, refs.p[u.ref1].cr };
        int lx[2], ly[2], cx[2], cy[2];
#pragma unroll
        for (int l = 0; l < 2; ++l) {
            int ax = mv[l][0], ay = mv[l][1];
            if (dmvr) clip_mv_dev(u.x, u.y, dst.w, dst.h, w, h, ax, ay);
            lx[l] = u.x + (ax >> 4) - 3;        ly[l] = u.y + (ay >> 4) - 3;
            cx[l] = (u.x >> 1) + (ax >> 5) - 1; cy[l] = (u.y >> 1) + (ay >> 5) - 1;
        }
        uint16_t *const lwin[2] = { s_wl[0] + 2 * XWIN_STRIDE + 4, s_wl[1] + 2 * XWIN_STRIDE + 4 };
        uint16_t *const cwin[4] = { s_wc[0][0] + 2 * XCWIN_STRIDE + 4, s_wc[0][1] + 2 * XCWIN_STRIDE + 4,
                                    s_wc[1][0] + 2 * XCWIN_STRIDE + 4, s_wc[1][1] + 2 * XCWIN_STRIDE + 4 };
        stage_unit_windows(dst, ry, rcb, rcr, lx, ly, cx, cy, w, h, 3, do_l, do_c, lane, lwin, XWIN_STRIDE, cwin, XCWIN_STRIDE, offl, offc);
#pragma unroll
        for (int l = 0; l < 2; ++l) {
            wl[l] = lwin[l] + offl[l];
            wcp[0][l] = cwin[l] + offc[l]; wcp[1][l] = cwin[2 + l] + offc[l];
        }
    }
    __syncthreads();

    // ---- 2. DMVR ----
    if (dmvr) {
        // 2a. apron (padd_dmvr / padd_dmvr_c): sides of the window rows, then whole rows
#pragma unroll
        for (int l = 0; l < 2; ++l) {
            if (do_l) pad_sides(wl[l], XWIN_STRIDE, w + 7, h + 7, lane);
            if (do_c) { pad_sides(wcp[0][l], XCWIN_STRIDE, wc + 3, hc + 3, lane); pad_sides(wcp[1][l], XCWIN_STRIDE, wc + 3, hc + 3, lane); }
        }
        __syncthreads();
#pragma unroll
        for (int l = 0; l < 2; ++l) {
            if (do_l) pad_rows(wl[l], XWIN_STRIDE, w + 7, h + 7, lane);
            if (do_c) { pad_rows(wcp[0][l], XCWIN_STRIDE, wc + 3, hc + 3, lane); pad_rows(wcp[1][l], XCWIN_STRIDE, wc + 3, hc + 3, lane); }
        }
        // 2b. bilinear blocks B_l[j][i], i, j = 0 .. w+3 / h+3 <-> sample offsets -2 .. : lane = (list, column).
        //     The right-hand sample of column i is the left-hand sample of column i + 1: one LDS read per row and
        //     a DPP wave shift instead of two reads (lane w+4 of each list only supplies that neighbour).
        {
            const int l = lane >= 32, i = lane & 31;
            if (i <= w + 4) {
                const int fx = ini[l][0] & 15, fy = ini[l][1] & 15;
                const uint16_t *src = wl[l] + XWIN_STRIDE + i + 1;            // offset -2 = window index 1
                int16_t *o = s_x[l] + i;
                int tp = 0;
                for (int j = 0; j < h + 5; ++j) {
                    const int a = src[j * XWIN_STRIDE];
                    const int b = __builtin_amdgcn_update_dpp(0, a, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
                    // (24-bit multiplies are full rate; v_mul_lo_u32 takes four issue slots)
                    const int t = fx ? (__mul24(16 - fx, a) + __mul24(fx, b) + 8) >> 4 : a;
                    if (j && i < w + 4) o[(j - 1) * BIL_STRIDE] = (int16_t)(fy ? (__mul24(16 - fy, tp) + __mul24(fy, t) + 8) >> 4 : tp);
                    tp = t;
                }
            }
        }
        __syncthreads();
        // 2c. SADs on every second row: lane = search point k (0..24) x row half.  Rows are read as packed dwords
        //     (odd start columns re-aligned with v_alignbit) and accumulated two samples at a time with v_sad_u16.
        int sad = 0;
        {
            const int k = lane < 25 ? lane : lane - 25, half = lane >= 25;
            if (lane < 50) {
                const int dx = k % 5 - 2, dy = k / 5 - 2;
                const int c0 = 2 + dx, c1 = 2 - dx;                       // same parity
                const bool odd = c0 & 1;
                const int *r0 = reinterpret_cast<const int *>(s_x[0] + (2 + dy) * BIL_STRIDE) + (c0 >> 1);
                const int *r1 = reinterpret_cast<const int *>(s_x[1] + (2 - dy) * BIL_STRIDE) + (c1 >> 1);
                const int nrow = h >> 2;                                   // rows per half (of the h/2 used)
                uint32_t acc = 0;
                for (int jj = 0; jj < nrow; ++jj) {
                    const int ro = 2 * (half * nrow + jj) * (BIL_STRIDE / 2);
                    if (w == 16) acc = sad_row<8>(r0 + ro, r1 + ro, odd, acc);
                    else         acc = sad_row<4>(r0 + ro, r1 + ro, odd, acc);
                }
                sad = (int)acc;
            }
            sad += __shfl(sad, lane + 25 < 64 ? lane + 25 : lane);
        }
        const int sad_c = __shfl(sad, 12);
        int min_cost = sad_c - (sad_c >> 2);
        if (min_cost >= w * h) {
            if (lane == 12) sad = min_cost;
            // arg-min with dmvr_compute_sads_*'s tie rule: lowest cost; the centre wins a tie, then the lowest index
            int key = lane < 25 ? (sad << 5) + (lane == 12 ? 0 : lane + 1) : 0x7fffffff;
#pragma unroll
            for (int m = 1; m < 32; m <<= 1) key = min(key, __shfl_xor(key, m));
            const int kk = key & 31, idx = kk ? kk - 1 : 12;
            int dh = (idx % 5 - 2) << 4, dv = (idx / 5 - 2) << 4;
            min_cost = key >> 5;
            const int s0 = min_cost;
            const int s1 = __shfl(sad, max(idx - 1, 0)), s3 = __shfl(sad, min(idx + 1, 24));
            const int s2 = __shfl(sad, max(idx - 5, 0)), s4 = __shfl(sad, min(idx + 5, 24));
            if (abs(dh) != 32 && abs(dv) != 32) {
                const int den_h = s1 + s3 - 2 * s0, den_v = s2 + s4 - 2 * s0;
                if (den_h) dh += (s1 != s0 && s3 != s0) ? div_for_maxq7((s1 - s3) << 4, den_h) : (s1 == s0 ? -8 : 8);
                if (den_v) dv += (s2 != s0 && s4 != s0) ? div_for_maxq7((s2 - s4) << 4, den_v) : (s2 == s0 ? -8 : 8);
            }
            dh = __builtin_amdgcn_readfirstlane(dh); dv = __builtin_amdgcn_readfirstlane(dv);
            mv[0][0] = ov_clip3(mv[0][0] + dh, -(1 << 17), (1 << 17) - 1); mv[0][1] = ov_clip3(mv[0][1] + dv, -(1 << 17), (1 << 17) - 1);
            mv[1][0] = ov_clip3(mv[1][0] - dh, -(1 << 17), (1 << 17) - 1); mv[1][1] = ov_clip3(mv[1][1] - dv, -(1 << 17), (1 << 17) - 1);
        }
        min_cost = __builtin_amdgcn_readfirstlane(min_cost);
        if (use_bdof && min_cost < 2 * w * h) use_bdof = false;
        __syncthreads();
    }
    if (mv_out && lane == 0) {
        int4 o; o.x = mv[0][0]; o.y = mv[0][1]; o.z = mv[1][0]; o.w = mv[1][1];
        *reinterpret_cast<int4 *>(mv_out + 4 * (size_t)bid) = o;
    }
    if (SEARCH_ONLY) return;

    // ---- 3. horizontal passes at the refined position: both lists (and both chroma planes) in one task loop ----
    const uint32_t *fvl[2];
    int ldx[2], ldy[2], cdx[2], cdy[2], ext[2][2];
    int fxs[2];
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        int fx = mv[l][0] & 15, fy = mv[l][1] & 15;
        if (u.flags & OVHIP_MC_HPEL_FILT) { if (fx == 8) fx = 16; if (fy == 8) fy = 16; }
        ext[l][0] = fx >= 8; ext[l][1] = fy >= 8;
        fvl[l] = g_taps.luma[fy];
        ldx[l] = (mv[l][0] >> 4) - (ini[l][0] >> 4); ldy[l] = (mv[l][1] >> 4) - (ini[l][1] >> 4);
        cdx[l] = (mv[l][0] >> 5) - (ini[l][0] >> 5); cdy[l] = (mv[l][1] >> 5) - (ini[l][1] >> 5);
        fxs[l] = fx;
    }
    // horizontal taps per lane (its window's list), as in k_mc2
    const int hfx = (lane >> 5) ? fxs[1] : fxs[0], hfc = ((lane >> 4) & 1 ? mv[1][0] : mv[0][0]) & 31;
    const uint4 hl_tp = *reinterpret_cast<const uint4 *>(g_taps.luma[hfx]);
    const uint2 hc_tp = *reinterpret_cast<const uint2 *>(g_taps.chroma[hfc]);
    // lanes dealt to the windows in fixed groups (luma 2 x 32, chroma 4 x 16), as in k_mc2: only the row changes in the loops
    if (do_l) {
        const int log2seg = log2w - 2;
        const int l = lane >> 5, tl = lane & 31;
        const int tp[4] = { (int)hl_tp.x, (int)hl_tp.y, (int)hl_tp.z, (int)hl_tp.w };
        const uint16_t *src = s_wl[0] + l * (XWIN_ROWS * XWIN_STRIDE) + (2 + (l ? ldy[1] : ldy[0])) * XWIN_STRIDE;
        const int off = 4 + (l ? offl[1] + ldx[1] : offl[0] + ldx[0]);
        const bool ident = hfx == 0;
        int16_t *ht = s_hl[0] + l * 16 * HT_STRIDE;
        const int x0 = (tl & ((1 << log2seg) - 1)) << 2, rstep = 32 >> log2seg;
        for (int r = tl >> log2seg; r < h + 7; r += rstep) h_task<8>(src + r * XWIN_STRIDE, off, x0, tp, ident, ht, HT_STRIDE, r);
    }
    if (do_c) {
        const int log2seg = log2wc > 2 ? log2wc - 2 : 0;
        const int qi = lane >> 4, tl = lane & 15, l = qi & 1;            // window qi = plane * 2 + list
        const int tp[2] = { (int)hc_tp.x, (int)hc_tp.y };
        const uint16_t *src = s_wc[0][0] + qi * (XCWIN_ROWS * XCWIN_STRIDE) + (2 + (l ? cdy[1] : cdy[0])) * XCWIN_STRIDE;
        const int off = 4 + (l ? offc[1] + cdx[1] : offc[0] + cdx[0]);
        const bool ident = hfc == 0;
        int16_t *ht = s_hc[0][0] + qi * 8 * CHT_STRIDE;
        const int x0 = (tl & ((1 << log2seg) - 1)) << 2, rstep = 16 >> log2seg;
        for (int r = tl >> log2seg; r < hc + 3; r += rstep) h_task<4>(src + r * XCWIN_STRIDE, off, x0, tp, ident, ht, CHT_STRIDE, r);
    }
    __syncthreads();

    // ---- 4. luma: vertical passes, then BDOF or the plain average ----
    if (do_l) {
        int P[2][4];
        v_pass<8>(s_hl[0], HT_STRIDE, log2w, h, fvl[0], lane, P[0]);
        v_pass<8>(s_hl[1], HT_STRIDE, log2w, h, fvl[1], lane, P[1]);
        const bool act = lane < ((h >> 2) << log2w);
        const int x = lane & (w - 1), g = lane >> log2w;
        int out[4];
        if (use_bdof) {
            // 4a. R tiles: interior = prediction, ring = integer reference samples << 4 (extend_bdof_buff)
            if (act) {
#pragma unroll
                for (int l = 0; l < 2; ++l)
#pragma unroll
                    for (int j = 0; j < 4; ++j) s_x[l][(4 * g + j + 1) * R_STRIDE + x + 1] = (int16_t)P[l][j];
            }
            const int nring = 2 * (w + 2) + 2 * h;
            for (int t = lane; t < 2 * nring; t += 64) {
                const int l = t >= nring, e = l ? t - nring : t;
                int i, j;
                if (e < w + 2)            { i = e; j = 0; }
                else if (e < 2 * (w + 2)) { i = e - (w + 2); j = h + 1; }
                else if (e < 2 * (w + 2) + h) { i = 0; j = e - 2 * (w + 2) + 1; }
                else                      { i = w + 1; j = e - 2 * (w + 2) - h + 1; }
                const uint16_t *sp = wl[l] + (3 + ldy[l] + j - 1 + ext[l][1]) * XWIN_STRIDE + 3 + ldx[l] + i - 1 + ext[l][0];
                s_x[l][j * R_STRIDE + i] = (int16_t)(*sp << 4);
            }
            __syncthreads();
            // 4b. gradients of the lane's 4 samples (compute_prof_grad), their list averages / differences
            int dgx[4], dgy[4];
            if (act) {
                int col[2][6];
#pragma unroll
                for (int l = 0; l < 2; ++l)
#pragma unroll
                    for (int j = 0; j < 6; ++j) col[l][j] = s_x[l][(4 * g + j) * R_STRIDE + x + 1] >> 6;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int o = (4 * g + j + 1) * R_STRIDE + x + 1;
                    const int gx0 = (s_x[0][o + 1] >> 6) - (s_x[0][o - 1] >> 6), gx1 = (s_x[1][o + 1] >> 6) - (s_x[1][o - 1] >> 6);
                    const int gy0 = col[0][j + 2] - col[0][j], gy1 = col[1][j + 2] - col[1][j];
                    dgx[j] = gx0 - gx1; dgy[j] = gy0 - gy1;
                    const int ax = (gx0 + gx1) >> 1, ay = (gy0 + gy1) >> 1, dr = (P[1][j] >> 4) - (P[0][j] >> 4);
                    // the five terms a 6x6 window sums, once per SAMPLE here instead of once per window that holds it
                    const int t_xy = ay < 0 ? -ax : (ay == 0 ? 0 : ax);
                    const int t_dx = ax < 0 ? -dr : (ax == 0 ? 0 : dr);
                    const int t_dy = ay < 0 ? -dr : (ay == 0 ? 0 : dr);
                    const int idx = (4 * g + j) * 16 + x;
                    s_ba[idx] = (uint32_t)abs(ax) | ((uint32_t)abs(ay) << 16);
                    s_bb[idx] = ((uint32_t)t_xy & 0xffffu) | ((uint32_t)t_dx << 16);
                    s_bc[idx] = (int16_t)t_dy;
                }
            }
            __syncthreads();
            // 4c. 6x6 window sums (derive_bdof_weights): the 4 lanes of a 4x4 block take 9 window samples each;
            //     gradients and predictions are replicated outside the block (extend_bdof_grad)
            int wx = 0, wy = 0;
            if (act) {
                // lane q of the block takes the half-rows 3q .. 3q+2 of the 6x6 window (3 samples each); a lane's 9 terms
                // fit int16 (|ax|, |ay| < 2^10, |dr| < 2^11), so the partial sums are packed adds
                typedef short bd_s2 __attribute__((ext_vector_type(2)));
                const int q = x & 3, sx = x & ~3, sy = 4 * g;
                bd_s2 a2 = (bd_s2)(0), b2 = (bd_s2)(0);
                int s_dy = 0;
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const int hr = 3 * q + t, r = hr >> 1, c0 = (hr & 1) * 3;
                    const int py = ov_clip3(sy - 1 + r, 0, h - 1);
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) {
                        const int px = ov_clip3(sx - 1 + c0 + cc, 0, w - 1), idx = py * 16 + px;
                        a2 += __builtin_bit_cast(bd_s2, s_ba[idx]);
                        b2 += __builtin_bit_cast(bd_s2, s_bb[idx]);
                        s_dy += s_bc[idx];
                    }
                }
                int s_ax = (unsigned short)a2.x, s_ay = (unsigned short)a2.y, s_xy = b2.x, s_dx = b2.y;
#pragma unroll
                for (int m = 1; m < 4; m <<= 1) {
                    s_ax += __shfl_xor(s_ax, m); s_ay += __shfl_xor(s_ay, m); s_xy += __shfl_xor(s_xy, m);
                    s_dx += __shfl_xor(s_dx, m); s_dy += __shfl_xor(s_dy, m);
                }
                if (s_ax) wx = ov_clip3((s_dx * 4) >> (31 - __clz(s_ax)), -15, 15);
                if (s_ay) {
                    const int x_off = wx ? (wx * s_xy) >> 1 : 0;
                    wy = ov_clip3(((s_dy * 4) - x_off) >> (31 - __clz(s_ay)), -15, 15);
                }
            }
            // 4d. rcn_apply_bdof_subblock
#pragma unroll
            for (int j = 0; j < 4; ++j) out[j] = ov_clip_bd((int)(int16_t)((P[0][j] + P[1][j] + __mul24(wx, dgx[j]) + __mul24(wy, dgy[j]) + 16) >> 5));
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) out[j] = ov_clip_bd((P[0][j] + P[1][j] + 16) >> 5);
        }
        if (act) {
            uint16_t *d = dst.y + ov_rowoff(u.y + 4 * g, dst.stride_y) + u.x + x;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int v = out[j];
                if ((u.flags & OVHIP_MC_LMCS) && lmcs_fwd) v = lmcs_fwd[v];
                d[j * dst.stride_y] = (uint16_t)v;
            }
        }
    }
    // ---- 5. chroma: plain average; both planes in one pass, NOUT rows per lane so that the lanes cover both ----
    if (do_c) {
        int tvc[2][2];                                   // vertical chroma taps, loaded where they are used
        load_taps<2>(g_taps.chroma[mv[0][1] & 31], tvc[0]); load_taps<2>(g_taps.chroma[mv[1][1] & 31], tvc[1]);
        if (wc * hc >= 64) chroma_avg<2>(u, dst, s_hc[0][0], tvc, lane, log2wc, hc);
        else               chroma_avg<1>(u, dst, s_hc[0][0], tvc, lane, log2wc, hc);
    }
    __syncthreads();          // LDS is reused by the next unit
    }
}

} // namespace


// =====================================================================================================
// K9: affine units.  One wavefront per <=16x16 luma area = up to 16 4x4 sub-blocks with their own motion
// vectors (+ up to 4 4x4 chroma blocks).  lane = (sub-block, column): the 4 lanes of a sub-block stage
// its 9x9 window (the 4x4-block filter ov_mc_filters_4 is 6-tap, rcn_mc.c:125-142), run the horizontal
// pass on 3 rows each, then each lane owns one output column: vertical pass, PROF gradient/refinement
// on a 6x6 LDS tile (extend_prof_buff / compute_prof_grad / rcn_prof, rcn_prof_bdof.c:152-290), bi / BCW
// combine, LMCS, store.  Replaces the per-sub-block calls of rcn_affine_mcp_b_l / _prof_mcp_b_l / _mcp_b_c
// (drv_affine_mvp.c:3264-3411).
// =====================================================================================================
namespace {

#define AWS 12     /* luma window row: <= 3 aligned qwords                     */
#define AHS 12     /* transposed H tile: 9 rows per column, 8-byte aligned     */
#define ACS 12     /* chroma window row: off(<=3) + 7 -> 3 qwords              */
#define ACHS 8     /* chroma transposed H tile: 7 rows per column              */

__device__ __forceinline__ int aff_combine(int dir, int w0, int w1, int p0, int p1)
{
    if (dir != 3)           return ov_clip_bd(((dir == 1 ? p0 : p1) + 8) >> 4);
    if (w0 == 4 && w1 == 4) return ov_clip_bd((p0 + p1 + 16) >> 5);
    return ov_clip_bd((__mul24(p1, w1) + __mul24(p0, w0) + 64) >> 7);
}

// window of ROWS x COLS samples -> LDS rows of 12 samples.  4 lanes (c = 0..3) per window.
template <int ROWS, int COLS>
__device__ __forceinline__ int aff_stage(const uint16_t *__restrict__ ref, int rstride, int rw, int rh, int sx0, int sy0,
                                         int c, uint16_t *win)
{
    const int ax = sx0 & ~3;
    int off = sx0 - ax;
    const int nq = (off + COLS + 3) >> 2;
    const bool fast = ax >= 0 && ax + 4 * nq <= rw && sy0 >= 0 && sy0 + ROWS <= rh && !(rstride & 3);
    if (fast) {
        if (c < nq) {
            uint2 q[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; ++r) q[r] = *reinterpret_cast<const uint2 *>(ref + ov_rowoff(sy0 + r, rstride) + ax + 4 * c);
#pragma unroll
            for (int r = 0; r < ROWS; ++r) *reinterpret_cast<uint2 *>(win + r * 12 + 4 * c) = q[r];
        }
    } else if (!((rstride | rw) & 3)) {
        // border, aligned geometry: clamped rows, outside groups = replicated edge sample (see WinStage)
        if (c < nq) {
            const int qx = ax + 4 * c, side = qx < 0 ? -1 : qx >= rw ? 1 : 0;
            const uint16_t *base = ref + ov_clip3(qx, 0, rw - 4);
            uint2 q[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; ++r) q[r] = *reinterpret_cast<const uint2 *>(base + ov_rowoff(ov_clip3(sy0 + r, 0, rh - 1), rstride));
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                uint2 v = q[r];
                if (side) { const uint32_t e = (side < 0 ? v.x & 0xffffu : v.y >> 16) * 0x10001u; v.x = e; v.y = e; }
                *reinterpret_cast<uint2 *>(win + r * 12 + 4 * c) = v;
            }
        }
    } else {
        off = 0;
        for (int i = c; i < COLS; i += 4) {
            const int sx = ov_clip3(sx0 + i, 0, rw - 1);
#pragma unroll 1
            for (int r = 0; r < ROWS; ++r) win[r * 12 + i] = ref[ov_clip3(sy0 + r, 0, rh - 1) * rstride + sx];
        }
    }
    return off;
}

#ifdef OV_MCA_PHASES
// Debug build only (-DOV_MCA_PHASES): per-unit shader-clock phase times of k_mca (tools/probe_mc_phases.py).
#define OV_MCA_PHASE_UNITS 65536
__device__ unsigned int g_mca_phase[OV_MCA_PHASE_UNITS * 8];
#define OV_APHASE(i) do { __builtin_amdgcn_s_waitcnt(0); const unsigned long long t_ = __builtin_readcyclecounter(); \
                          ph[i] = (unsigned int)(t_ - tprev); tprev = t_; } while (0)
#else
#define OV_APHASE(i) do { } while (0)
#endif

// Luma window of one sub-block and list, split into issue (loads into registers) and park (registers -> LDS) so that
// list 1's loads are in flight while list 0 is being filtered.  Geometry that is not a multiple of 4 samples is
// fetched sample by sample with clamped coordinates at park time.
struct AffLumaStage {
    uint2 q[9];
    const uint16_t *ref;
    int sx0, sy0, off, side;
    bool fast;

    __device__ __forceinline__ void issue(const uint16_t *__restrict__ r, int rstride, int rw, int rh, int x0, int y0, int c)
    {
        ref = r; sx0 = x0; sy0 = y0;
        const int ax = sx0 & ~3;
        off = sx0 - ax;
        const int nq = (off + 9 + 3) >> 2;
        fast = !((rstride | rw) & 3);
        if (fast && c < nq) {
            // rows clamp per lane; an aligned group outside the picture = the edge sample replicated (see WinStage)
            const int qx = ax + 4 * c;
            side = qx < 0 ? -1 : qx >= rw ? 1 : 0;
            const uint16_t *base = ref + ov_clip3(qx, 0, rw - 4);
#pragma unroll
            for (int k = 0; k < 9; ++k) q[k] = *reinterpret_cast<const uint2 *>(base + ov_rowoff(ov_clip3(sy0 + k, 0, rh - 1), rstride));
        }
    }
    __device__ __forceinline__ int park(int rstride, int rw, int rh, int c, uint16_t *win)
    {
        if (fast) {
            if (c < ((off + 9 + 3) >> 2)) {
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    uint2 v = q[k];
                    if (side) { const uint32_t e = (side < 0 ? v.x & 0xffffu : v.y >> 16) * 0x10001u; v.x = e; v.y = e; }
                    *reinterpret_cast<uint2 *>(win + k * AWS + 4 * c) = v;
                }
            }
            return off;
        }
        for (int i = c; i < 9; i += 4) {
            const int sx = ov_clip3(sx0 + i, 0, rw - 1);
#pragma unroll 1
            for (int k = 0; k < 9; ++k) win[k * AWS + i] = ref[ov_clip3(sy0 + k, 0, rh - 1) * rstride + sx];
        }
        return 0;
    }
};

#define MCA_LDS_WIN  (16 * 9 * AWS * 2)
#define MCA_LDS_HT   (16 * 4 * AHS * 2)
#define MCA_LDS_T    (16 * 40 * 2)
#define MCA_LDS_CWIN (2 * 8 * 7 * ACS * 2)
#define MCA_LDS      (MCA_LDS_WIN + MCA_LDS_HT + MCA_LDS_T + MCA_LDS_CWIN + 2 * 8 * 4 * ACHS * 2)

// units wg0, wg0 + wstride, ... (one single-wave workgroup; `lds` = MCA_LDS bytes, 16-byte aligned)
__device__ __forceinline__ void mca_units(const ovhip_pic &dst, const RefTable &refs, const ovhip_aff_unit *__restrict__ units,
                                          uint32_t n_units, const int32_t *__restrict__ side, const uint16_t *__restrict__ lmcs_fwd,
                                          uint32_t wg0, uint32_t wstride, char *lds)
{
    // Luma tiles hold ONE list at a time (list 1 is filtered after list 0, its window waiting in registers): with both
    // lists resident the 16 KB workgroup allowed 10 per CU and a 4K picture's affine units needed two rounds.
    uint16_t (*const s_win)[9 * AWS] = reinterpret_cast<uint16_t (*)[9 * AWS]>(lds);
    int16_t  (*const s_ht)[4 * AHS]  = reinterpret_cast<int16_t (*)[4 * AHS]>(lds + MCA_LDS_WIN);
    int16_t  (*const s_t)[40]        = reinterpret_cast<int16_t (*)[40]>(lds + MCA_LDS_WIN + MCA_LDS_HT);
    uint16_t (*const s_cwin)[8][7 * ACS] = reinterpret_cast<uint16_t (*)[8][7 * ACS]>(lds + MCA_LDS_WIN + MCA_LDS_HT + MCA_LDS_T);   // [list][comp * 4 + block]
    int16_t  (*const s_cht)[8][4 * ACHS] = reinterpret_cast<int16_t (*)[8][4 * ACHS]>(lds + MCA_LDS_WIN + MCA_LDS_HT + MCA_LDS_T + MCA_LDS_CWIN);

    const int lane = threadIdx.x;
    {   // one unit per workgroup
    const uint32_t wg = wg0;
    if (wg >= n_units) return;
    const uint32_t bid = wstride >= n_units ? ov_xcd_slot(wg, n_units) : wg;        // XCD-aware order, see k_mc2
#ifdef OV_MCA_PHASES
    unsigned int ph[8] = {}; unsigned long long tprev = __builtin_readcyclecounter();
#endif
    const ovhip_aff_unit u = units[bid];
    OV_APHASE(0);
    const int nsx = u.w >> 2, nsb = nsx * (u.h >> 2), ncx = u.w >> 3, ncb = ncx * (u.h >> 3);
    const bool do_c = !(u.flags & OVHIP_AFF_NO_CHROMA);
    const int4 *mvs = reinterpret_cast<const int4 *>(side + u.side_off);

    // ---- luma lane state: sub-block sb, column / window slice c ----
    const int sb = lane >> 2, c = lane & 3;
    const bool act = sb < nsb;
    const int bx = u.x + 4 * (sb % nsx), by = u.y + 4 * (sb / nsx);
    int4 m = make_int4(0, 0, 0, 0);
    if (act) m = mvs[sb];
    // ---- chroma lane state: list cl, window cw = comp * 4 + block, slice cc ----
    const int cl = lane >> 5, cwi = (lane >> 2) & 7, cblk = cwi & 3, ccomp = cwi >> 2, cc = lane & 3;
    const bool cact = do_c && cblk < ncb && (u.dir & (1 << cl));
    int4 cm = make_int4(0, 0, 0, 0);
    if (do_c && cblk < ncb) cm = mvs[nsb + cblk];
    const int cbx = (u.x >> 1) + 4 * (cblk % ncx), cby = (u.y >> 1) + 4 * (cblk / ncx);
    int coff = 0;

    OV_APHASE(1);

    // ---- 1. windows: the first luma list into registers, chroma straight to LDS; PROF offsets prefetched ----
    AffLumaStage st;
    const int rsy = dst.stride_y, rpw = dst.w, rph = dst.h;               // reference geometry = dst's (checked at launch)
    const int lfirst = (u.dir & 1) ? 0 : 1;
    if (act) {
        const int mvx = lfirst ? m.z : m.x, mvy = lfirst ? m.w : m.y;
        st.issue(refs.p[lfirst ? u.ref1 : u.ref0].y, rsy, rpw, rph, bx + (mvx >> 4) - 2, by + (mvy >> 4) - 2, c);
    }
    if (cact) {
        const ovhip_pic &rp = refs.p[cl ? u.ref1 : u.ref0];
        const int mvx = cl ? cm.z : cm.x, mvy = cl ? cm.w : cm.y;
        coff = aff_stage<7, 7>(ccomp ? rp.cr : rp.cb, rp.stride_c, rp.w >> 1, rp.h >> 1, cbx + (mvx >> 5) - 1, cby + (mvy >> 5) - 1, cc,
                               s_cwin[cl][cwi]);
    }
    bool prof[2];
    int pdx[2][4], pdy[2][4];
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        prof[l] = act && (u.dir & (1 << l)) && (u.flags & OVHIP_AFF_PROF) && (u.dir != 3 || ((u.prof_dir >> l) & 1));
        if (prof[l]) {
            const int16_t *pt = reinterpret_cast<const int16_t *>(side + u.prof_off) + 32 * l;
#pragma unroll
            for (int j = 0; j < 4; ++j) { pdx[l][j] = pt[4 * j + c]; pdy[l][j] = pt[16 + 4 * j + c]; }
        }
    }
    OV_APHASE(2);

    // ---- 2. luma, one list after the other: park, horizontal pass (rows c, c+4, c+8 of the sub-block's window,
    //         4 outputs each), vertical pass, PROF.  The chroma horizontal pass rides along with list 0's. ----
    int P[2][4] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 } };
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        const bool on = (u.dir & (1 << l)) && act;
        const int mvx = l ? m.z : m.x, mvy = l ? m.w : m.y;
        int off = 0;
        if (l) __syncthreads();                         // list 0's tiles are dead
        if (on) off = st.park(rsy, rpw, rph, c, s_win[sb]);
        if (l == 0 && u.dir == 3 && act)                // list 1's window flies while list 0 is filtered
            st.issue(refs.p[u.ref1].y, rsy, rpw, rph, bx + (m.z >> 4) - 2, by + (m.w >> 4) - 2, c);
        __syncthreads();
        if (on) {
            int tp[3];
            load_taps<3>(g_taps.luma6[mvx & 15], tp);
            for (int r = c; r < 9; r += 4) {
                int d[5], out[4];
                load_row_at<6>(s_win[sb] + r * AWS, off, d);
                fir4<6>(d, tp, out);
#pragma unroll
                for (int o = 0; o < 4; ++o) s_ht[sb][o * AHS + r] = (int16_t)(out[o] >> (OV_BD - 8));
            }
        }
        if (l == 0 && cact) {
            const int cmvx = cl ? cm.z : cm.x;
            int tp[2];
            load_taps<2>(g_taps.chroma[cmvx & 31], tp);
            for (int r = cc; r < 7; r += 4) {
                int d[4], out[4];
                load_row_at<4>(s_cwin[cl][cwi] + r * ACS, coff, d);
                fir4<4>(d, tp, out);
#pragma unroll
                for (int o = 0; o < 4; ++o) s_cht[cl][cwi][o * ACHS + r] = (int16_t)(out[o] >> (OV_BD - 8));
            }
        }
        __syncthreads();
        if (on) {
            int tp[3], d[5];
            load_taps<3>(g_taps.luma6[mvy & 15], tp);
            const int *q = reinterpret_cast<const int *>(s_ht[sb] + c * AHS);
#pragma unroll
            for (int j = 0; j < 5; ++j) d[j] = q[j];
            fir4<6>(d, tp, P[l]);
#pragma unroll
            for (int o = 0; o < 4; ++o) P[l][o] >>= 6;
        }
        if (prof[l]) {
            // 6x6 tile: interior = prediction column, ring = integer reference samples << 4 (5 ring samples per lane)
            int16_t *t = s_t[sb];
            const int ex = (mvx & 15) >> 3, ey = (mvy & 15) >> 3;
#pragma unroll
            for (int j = 0; j < 4; ++j) t[(j + 1) * 6 + c + 1] = (int16_t)P[l][j];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int e = 5 * c + k;
                int i, j;
                if (e < 6)       { i = e; j = 0; }
                else if (e < 12) { i = e - 6; j = 5; }
                else if (e < 16) { i = 0; j = e - 11; }
                else             { i = 5; j = e - 15; }
                t[j * 6 + i] = (int16_t)(s_win[sb][(j + 1 + ey) * AWS + off + i + 1 + ex] << 4);
            }
        }
        __syncthreads();
        if (prof[l]) {
            const int16_t *t = s_t[sb];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int o = (j + 1) * 6 + c + 1;
                const int gx = (t[o + 1] >> 6) - (t[o - 1] >> 6), gy = (t[o + 6] >> 6) - (t[o - 6] >> 6);
                const int add = ov_clip3(__mul24(pdx[l][j], gx) + __mul24(pdy[l][j], gy), -(1 << 13), (1 << 13) - 1);
                P[l][j] = (int)(int16_t)(P[l][j] + add);
            }
        }
    }
    OV_APHASE(4);
    if (act) {
        const int dir = ((u.ident_l >> sb) & 1) ? 2 : u.dir;
        uint16_t *d = dst.y + ov_rowoff(by, dst.stride_y) + bx + c;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int v = aff_combine(dir, u.w0, u.w1, P[0][j], P[1][j]);
            if ((u.flags & OVHIP_AFF_LMCS) && lmcs_fwd) v = lmcs_fwd[v];
            d[j * dst.stride_y] = (uint16_t)v;
        }
    }
    OV_APHASE(5);
    // ---- 4. chroma: lanes 0..31 = (comp, block, column), both lists ----
    if (do_c && lane < 32 && cblk < ncb) {
        int Pc[2][4] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 } };
#pragma unroll
        for (int l = 0; l < 2; ++l) {
            if (!(u.dir & (1 << l))) continue;
            const int mvy = l ? cm.w : cm.y;
            int tp[2], d[4];
            load_taps<2>(g_taps.chroma[mvy & 31], tp);
            const int *q = reinterpret_cast<const int *>(s_cht[l][cwi] + cc * ACHS);
#pragma unroll
            for (int j = 0; j < 4; ++j) d[j] = q[j];
            fir4<4>(d, tp, Pc[l]);
#pragma unroll
            for (int o = 0; o < 4; ++o) Pc[l][o] >>= 6;
        }
        const int dir = ((u.ident_c >> cblk) & 1) ? 2 : u.dir;
        uint16_t *d = (ccomp ? dst.cr : dst.cb) + ov_rowoff(cby, dst.stride_c) + cbx + cc;
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j * dst.stride_c] = (uint16_t)aff_combine(dir, u.w0, u.w1, Pc[0][j], Pc[1][j]);
    }
    __syncthreads();
#ifdef OV_MCA_PHASES
    OV_APHASE(6);
    if (lane == 0 && bid < OV_MCA_PHASE_UNITS) {
#pragma unroll
        for (int i = 0; i < 7; ++i) g_mca_phase[bid * 8 + i] = ph[i];
        g_mca_phase[bid * 8 + 7] = 1;
    }
#endif
    }
}

__global__ __launch_bounds__(64) OV_OCC_MCX void k_mcx(ovhip_pic dst, RefTable refs, const ovhip_mc_unit *__restrict__ units,
                                             uint32_t n_units, const uint16_t *__restrict__ lmcs_fwd, int32_t *__restrict__ mv_out)
{
    __shared__ __attribute__((aligned(16))) char lds[MCX_LDS];
    mcx_units<false>(dst, refs, units, n_units, lmcs_fwd, mv_out, blockIdx.x, gridDim.x, lds);
}

__global__ __launch_bounds__(64) OV_OCC_MCX void k_dmvr_search(ovhip_pic geom, RefTable refs, const ovhip_mc_unit *__restrict__ units,
                                                     uint32_t n_units, int32_t *__restrict__ mv_out)
{
    __shared__ __attribute__((aligned(16))) char lds[MCX_LDS];
    mcx_units<true>(geom, refs, units, n_units, nullptr, mv_out, blockIdx.x, gridDim.x, lds);
}

__global__ __launch_bounds__(64) OV_OCC_MCX void k_mca(ovhip_pic dst, RefTable refs, const ovhip_aff_unit *__restrict__ units,
                                             uint32_t n_units, const int32_t *__restrict__ side, const uint16_t *__restrict__ lmcs_fwd)
{
    __shared__ __attribute__((aligned(16))) char lds[MCA_LDS];
    mca_units(dst, refs, units, n_units, side, lmcs_fwd, blockIdx.x, gridDim.x, lds);
}

// Both kinds of unit in ONE launch: workgroups [0, n_a) take the affine units (few, long, latency-bound -- first, so that
// the refined units hide them), [n_a, n_a + n_x) the BDOF / DMVR units.  One kernel boundary and one tail less.
__global__ __launch_bounds__(64) OV_OCC_MCX void k_mcxa(ovhip_pic dst, RefTable refs, const ovhip_mc_unit *__restrict__ xunits, uint32_t n_x,
                                              int32_t *__restrict__ mv_out, const ovhip_aff_unit *__restrict__ aunits, uint32_t n_a,
                                              const int32_t *__restrict__ side, const uint16_t *__restrict__ lmcs_fwd)
{
    __shared__ __attribute__((aligned(16))) char lds[MCX_LDS > MCA_LDS ? MCX_LDS : MCA_LDS];
    if (blockIdx.x < n_a) mca_units(dst, refs, aunits, n_a, side, lmcs_fwd, blockIdx.x, n_a, lds);
    else                  mcx_units<false>(dst, refs, xunits, n_x, lmcs_fwd, mv_out, blockIdx.x - n_a, n_x, lds);
}

} // namespace

#ifdef OV_MCA_PHASES
extern "C" int ovhip_debug_mca_phases(unsigned int *out /* [65536][8] */)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mca_phase), sizeof(unsigned int) * OV_MCA_PHASE_UNITS * 8) == hipSuccess ? OVHIP_OK : OVHIP_ELAUNCH;
}
#endif

extern "C" int ovhip_mcx_launch(ovhip_ctx *ctx, const ovhip_pic *dst, const ovhip_pic *refs, uint32_t n_refs,
                                const ovhip_mc_unit *d_units, uint32_t n_units, const uint16_t *d_lmcs_fwd_lut,
                                int32_t *d_mv_out)
{
    if (!ctx || !dst) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    if (!n_units) return OVHIP_OK;
    if (!refs || !n_refs || n_refs > MC_MAX_REFS || !d_units)
        return ov_fail(ctx, OVHIP_EINVAL, "ovhip_mcx_launch: bad reference table / units", hipSuccess);
    RefTable t;
    memset(&t, 0, sizeof(t));
    for (uint32_t i = 0; i < n_refs; ++i) {
        if (refs[i].w != dst->w || refs[i].h != dst->h || refs[i].stride_y != dst->stride_y || refs[i].stride_c != dst->stride_c)
            return ov_fail(ctx, OVHIP_EUNSUP, "ovhip_mcx_launch: reference picture geometry differs from dst (RPR)", hipSuccess);
        t.p[i] = refs[i];
    }
    for (uint32_t i = n_refs; i < MC_MAX_REFS; ++i) t.p[i] = refs[0];
    hipLaunchKernelGGL(k_mcx, dim3(n_units), dim3(64), 0, ctx->stream, *dst, t, d_units, n_units, d_lmcs_fwd_lut, d_mv_out);
    OV_LAUNCH_CHECK(ctx, "k_mcx");
    return OVHIP_OK;
}

extern "C" int ovhip_dmvr_search_launch(ovhip_ctx *ctx, const ovhip_pic *geom, const ovhip_pic *refs, uint32_t n_refs,
                                        const ovhip_mc_unit *d_units, uint32_t n_units, int32_t *d_mv_out)
{
    if (!ctx || !geom) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    if (!n_units) return OVHIP_OK;
    if (!refs || !n_refs || n_refs > MC_MAX_REFS || !d_units || !d_mv_out)
        return ov_fail(ctx, OVHIP_EINVAL, "ovhip_dmvr_search_launch: bad reference table / units", hipSuccess);
    RefTable t;
    memset(&t, 0, sizeof(t));
    for (uint32_t i = 0; i < n_refs; ++i) {
        if (refs[i].w != geom->w || refs[i].h != geom->h || refs[i].stride_y != geom->stride_y || refs[i].stride_c != geom->stride_c)
            return ov_fail(ctx, OVHIP_EUNSUP, "ovhip_dmvr_search_launch: reference picture geometry differs (RPR)", hipSuccess);
        t.p[i] = refs[i];
    }
    for (uint32_t i = n_refs; i < MC_MAX_REFS; ++i) t.p[i] = refs[0];
    hipLaunchKernelGGL(k_dmvr_search, dim3(n_units), dim3(64), 0, ctx->stream, *geom, t, d_units, n_units, d_mv_out);
    OV_LAUNCH_CHECK(ctx, "k_dmvr_search");
    return OVHIP_OK;
}

extern "C" int ovhip_mca_launch(ovhip_ctx *ctx, const ovhip_pic *dst, const ovhip_pic *refs, uint32_t n_refs,
                                const ovhip_aff_unit *d_units, uint32_t n_units, const int32_t *d_side,
                                const uint16_t *d_lmcs_fwd_lut)
{
    if (!ctx || !dst) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    if (!n_units) return OVHIP_OK;
    if (!refs || !n_refs || n_refs > MC_MAX_REFS || !d_units || !d_side)
        return ov_fail(ctx, OVHIP_EINVAL, "ovhip_mca_launch: bad reference table / units / side arena", hipSuccess);
    RefTable t;
    memset(&t, 0, sizeof(t));
    for (uint32_t i = 0; i < n_refs; ++i) {
        if (refs[i].w != dst->w || refs[i].h != dst->h || refs[i].stride_y != dst->stride_y || refs[i].stride_c != dst->stride_c)
            return ov_fail(ctx, OVHIP_EUNSUP, "ovhip_mca_launch: reference picture geometry differs from dst (RPR)", hipSuccess);
        t.p[i] = refs[i];
    }
    for (uint32_t i = n_refs; i < MC_MAX_REFS; ++i) t.p[i] = refs[0];
    hipLaunchKernelGGL(k_mca, dim3(n_units), dim3(64), 0, ctx->stream, *dst, t, d_units, n_units, d_side, d_lmcs_fwd_lut);
    OV_LAUNCH_CHECK(ctx, "k_mca");
    return OVHIP_OK;
}

extern "C" int ovhip_mcxa_launch(ovhip_ctx *ctx, const ovhip_pic *dst, const ovhip_pic *refs, uint32_t n_refs,
                                 const ovhip_mc_unit *d_xunits, uint32_t n_xunits, int32_t *d_mv_out,
                                 const ovhip_aff_unit *d_aunits, uint32_t n_aunits, const int32_t *d_side,
                                 const uint16_t *d_lmcs_fwd_lut)
{
    if (!ctx || !dst) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    if (!n_aunits) return ovhip_mcx_launch(ctx, dst, refs, n_refs, d_xunits, n_xunits, d_lmcs_fwd_lut, d_mv_out);
    if (!n_xunits) return ovhip_mca_launch(ctx, dst, refs, n_refs, d_aunits, n_aunits, d_side, d_lmcs_fwd_lut);
    if (!refs || !n_refs || n_refs > MC_MAX_REFS || !d_xunits || !d_aunits || !d_side)
        return ov_fail(ctx, OVHIP_EINVAL, "ovhip_mcxa_launch: bad reference table / units / side arena", hipSuccess);
    RefTable t;
    memset(&t, 0, sizeof(t));
    for (uint32_t i = 0; i < n_refs; ++i) {
        if (refs[i].w != dst->w || refs[i].h != dst->h || refs[i].stride_y != dst->stride_y || refs[i].stride_c != dst->stride_c)
            return ov_fail(ctx, OVHIP_EUNSUP, "ovhip_mcxa_launch: reference picture geometry differs from dst (RPR)", hipSuccess);
        t.p[i] = refs[i];
    }
    for (uint32_t i = n_refs; i < MC_MAX_REFS; ++i) t.p[i] = refs[0];
    hipLaunchKernelGGL(k_mcxa, dim3(n_aunits + n_xunits), dim3(64), 0, ctx->stream, *dst, t, d_xunits, n_xunits, d_mv_out,
                       d_aunits, n_aunits, d_side, d_lmcs_fwd_lut);
    OV_LAUNCH_CHECK(ctx, "k_mcxa");
    return OVHIP_OK;
}
