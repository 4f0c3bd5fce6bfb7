/* ovvc_oracle_intra.c -- TEST INFRASTRUCTURE: CPU restatement of OpenVVC's intra prediction and of the ordered
 * ("intra") pass of the device engine.  Included by ovvc_oracle.c; the header there applies: nothing under oracle/ is
 * linked, loaded or called by the product.
 *
 * Restates, task by task (ovhip_itask, include/ovvc_hip.h):
 *   reference-sample fetch + substitution        libovvc/rcn_fill_ref.c:72-600  (H.266 8.4.5.2.8-9)
 *   [1 2 1] reference smoothing                   rcn_fill_ref.c:41-68           (8.4.5.2.10)
 *   planar / DC (+ PDPC)                          rcn_intra_dc_planar.c          (8.4.5.2.11-12, .15)
 *   angular incl. wide angles, fC / fG, PDPC      rcn_intra.c:46-483, rcn_intra_angular.c (8.4.5.2.13, .15)
 *   multi-reference-line prediction               rcn_intra.c:646-770
 *   chroma: planar / DC / angular, CCLM / MDLM    rcn_intra.c:773-1180, rcn_intra_cclm.c:56-880 (8.4.5.2.14)
 *   matrix-based intra prediction                 rcn_intra_mip.c:44-400 (8.4.5.2.2-5)
 *   block-DPCM prediction (pure H / V copy)       rcn_intra.c:511-523, :1044-1075
 *   CIIP's planar + blend                         rcn_inter.c:2968-3067
 * Pinned by tests/golden/intra.ovg (the reference's intra_pred / intra_pred_c / intra_pred_mrl / mip / cclm slots).
 */
#include "vvc_mip_tables.h"

typedef struct oracle_res { const int16_t *y, *cb, *cr; int32_t stride_y, stride_c; } oracle_res;

static const int16_t ang_table[32] = { 0, 1, 2, 3, 4, 6, 8, 10, 12, 14, 16, 18, 20, 23, 26, 29, 32, 35, 39, 45, 51, 57, 64, 73, 86, 102, 128, 171, 256, 341, 512, 1024 };
static const int16_t inv_ang_table[32] = { 0, 16384, 8192, 5461, 4096, 2731, 2048, 1638, 1365, 1170, 1024, 910, 819, 712, 630, 565, 512, 468, 420, 364,
                                           321, 287, 256, 224, 191, 161, 128, 96, 64, 48, 32, 16 };
static const uint8_t hv_dist_thres[8] = { 24, 24, 24, 14, 2, 0, 0, 0 };   /* intra_filter[], data_rcn_angular.c:48 */

static const int8_t fc_taps[32][4] = {     /* cubic interpolation filter, H.266 Table 25 */
    { 0, 64, 0, 0 }, { -1, 63, 2, 0 }, { -2, 62, 4, 0 }, { -2, 60, 7, -1 }, { -2, 58, 10, -2 }, { -3, 57, 12, -2 }, { -4, 56, 14, -2 },
    { -4, 55, 15, -2 }, { -4, 54, 16, -2 }, { -5, 53, 18, -2 }, { -6, 52, 20, -2 }, { -6, 49, 24, -3 }, { -6, 46, 28, -4 }, { -5, 44, 29, -4 },
    { -4, 42, 30, -4 }, { -4, 39, 33, -4 }, { -4, 36, 36, -4 }, { -4, 33, 39, -4 }, { -4, 30, 42, -4 }, { -4, 29, 44, -5 }, { -4, 28, 46, -6 },
    { -3, 24, 49, -6 }, { -2, 20, 52, -6 }, { -2, 18, 53, -5 }, { -2, 16, 54, -4 }, { -2, 15, 55, -4 }, { -2, 14, 56, -4 }, { -2, 12, 57, -3 },
    { -2, 10, 58, -2 }, { -1, 7, 60, -2 }, { 0, 4, 62, -2 }, { 0, 2, 63, -1 } };
static const int8_t fg_taps[32][4] = {     /* Gaussian interpolation filter */
    { 16, 32, 16, 0 }, { 16, 32, 16, 0 }, { 15, 31, 17, 1 }, { 15, 31, 17, 1 }, { 14, 30, 18, 2 }, { 14, 30, 18, 2 }, { 13, 29, 19, 3 }, { 13, 29, 19, 3 },
    { 12, 28, 20, 4 }, { 12, 28, 20, 4 }, { 11, 27, 21, 5 }, { 11, 27, 21, 5 }, { 10, 26, 22, 6 }, { 10, 26, 22, 6 }, { 9, 25, 23, 7 }, { 9, 25, 23, 7 },
    { 8, 24, 24, 8 }, { 8, 24, 24, 8 }, { 7, 23, 25, 9 }, { 7, 23, 25, 9 }, { 6, 22, 26, 10 }, { 6, 22, 26, 10 }, { 5, 21, 27, 11 }, { 5, 21, 27, 11 },
    { 4, 20, 28, 12 }, { 4, 20, 28, 12 }, { 3, 19, 29, 13 }, { 3, 19, 29, 13 }, { 2, 18, 30, 14 }, { 2, 18, 30, 14 }, { 1, 17, 31, 15 }, { 1, 17, 31, 15 } };

/* PDPC weight 32 >> ((i << 1) >> scale), 0 once the shift reaches the word size (vvc_pdpc_w[][], rcn_intra_dc_planar.c:42-60) */
static int pdpc_wgt(int i, int scale) { const int sh = (i << 1) >> scale; return sh > 5 ? 0 : 32 >> sh; }

static int ilog2(int v) { int n = -1; while (v > 0) { ++n; v >>= 1; } return n; }

#define REF_PAD 160
typedef struct { uint16_t a[REF_PAD + 2 * 128 + 8 + REF_PAD], l[REF_PAD + 2 * 128 + 8 + REF_PAD]; } ref_bufs;

/* Reference samples of a w x h block at (x0, y0) of `plane` on reference line `mrl`:
 *   abv[k] = p(x0 - 1 - mrl + k, y0 - 1 - mrl),  lft[k] = p(x0 - 1 - mrl, y0 - 1 - mrl + k),  k = 0 .. 2w (2h) + mrl
 * `unit` = samples per availability unit (4 luma, 2 chroma); corner = the above-left unit is available; avl_abv / avl_lft =
 * available units right of / below the corner.  Not-available samples are substituted as 8.4.5.2.8 does (walk from the bottom
 * of the left column up and then along the top row, copying the last available sample); with nothing available: 1 << 9.
 * Both arrays are extended by their last sample (wide angles, rcn_fill_ref.c "padding for wide angle"). */
static void
fetch_refs(const uint16_t *plane, int stride, int x0, int y0, int w, int h, int unit, int corner, int avl_abv, int avl_lft, int mrl,
           uint16_t *abv, uint16_t *lft)
{
    const int na = 2 * w + mrl + 1, nl = 2 * h + mrl + 1;
    const int xa = x0 - 1 - mrl, ya = y0 - 1 - mrl;
    uint8_t va[2 * 128 + 8], vl[2 * 128 + 8];
    for (int k = 0; k < na; ++k) {
        const int ok = k <= mrl ? corner : ((k - mrl - 1) / unit) < avl_abv;
        va[k] = (uint8_t)ok;
        abv[k] = ok ? plane[ya * stride + xa + k] : 0;
    }
    for (int k = 0; k < nl; ++k) {
        const int ok = k <= mrl ? corner : ((k - mrl - 1) / unit) < avl_lft;
        vl[k] = (uint8_t)ok;
        lft[k] = ok ? plane[(ya + k) * stride + xa] : 0;
    }
    int any = 0;
    for (int k = 0; k < na; ++k) any |= va[k];
    for (int k = 0; k < nl; ++k) any |= vl[k];
    if (!any) {
        for (int k = 0; k < na; ++k) abv[k] = 1 << (BD - 1);
        for (int k = 0; k < nl; ++k) lft[k] = 1 << (BD - 1);
    } else {
        /* order: lft[nl-1] .. lft[0] (= abv[0]), abv[1] .. abv[na-1] */
        if (!vl[nl - 1]) {
            uint16_t v = 0; int found = 0;
            for (int k = nl - 1; k >= 0 && !found; --k) if (vl[k]) { v = lft[k]; found = 1; }
            for (int k = 1; k < na && !found; ++k) if (va[k]) { v = abv[k]; found = 1; }
            lft[nl - 1] = v; vl[nl - 1] = 1;
        }
        for (int k = nl - 2; k >= 0; --k) if (!vl[k]) lft[k] = lft[k + 1];
        abv[0] = lft[0];
        for (int k = 1; k < na; ++k) if (!va[k]) abv[k] = abv[k - 1];
        /* reference quirk (rcn_fill_ref.c:353-357, :446-450): with the corner missing but both arms present (a slice starting at
         * the CTU above) the ABOVE array takes its corner from its own first sample, the left array from its own */
        if (!corner && mrl == 0 && va[1] && vl[1]) abv[0] = abv[1];
    }
    for (int k = na; k < na + REF_PAD; ++k) abv[k] = abv[na - 1];
    for (int k = nl; k < nl + REF_PAD; ++k) lft[k] = lft[nl - 1];
}

/* filter_ref_samples (rcn_fill_ref.c:41-68): dst[0] from the two arms, [1 2 1] / 4 up to len - 1, dst[len] copied */
static void
smooth_refs(const uint16_t *src, const uint16_t *other, uint16_t *dst, int len)
{
    dst[0] = (uint16_t)((other[1] + 2 * src[0] + src[1] + 2) >> 2);
    for (int i = 1; i < len; ++i) dst[i] = (uint16_t)((src[i + 1] + 2 * src[i] + src[i - 1] + 2) >> 2);
    for (int i = len; i < len + REF_PAD; ++i) dst[i] = src[i];
}

static void
pred_planar(const uint16_t *abv, const uint16_t *lft, int l2w, int l2h, int pdpc, uint16_t *out /* [h][w] */)
{
    const int w = 1 << l2w, h = 1 << l2h, scale = (l2w + l2h - 2) >> 2;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int pv = ((h - 1 - y) * abv[1 + x] + (y + 1) * lft[1 + h]) << l2w;
            const int ph = ((w - 1 - x) * lft[1 + y] + (x + 1) * abv[1 + w]) << l2h;
            int v = (pv + ph + (w * h)) >> (l2w + l2h + 1);
            if (pdpc) {
                const int wt = pdpc_wgt(y, scale), wl = pdpc_wgt(x, scale);
                v = clip_bd((lft[1 + y] * wl + abv[1 + x] * wt + (64 - wl - wt) * v + 32) >> 6);
            }
            out[y * w + x] = (uint16_t)v;
        }
}

static void
pred_dc(const uint16_t *abv, const uint16_t *lft, int l2w, int l2h, int pdpc, uint16_t *out)
{
    const int w = 1 << l2w, h = 1 << l2h, scale = (l2w + l2h - 2) >> 2;
    int sum = 0, dc;
    if (w >= h) for (int x = 0; x < w; ++x) sum += abv[1 + x];
    if (h >= w) for (int y = 0; y < h; ++y) sum += lft[1 + y];
    if (w == h) dc = (sum + w) >> (l2w + 1);
    else if (w > h) dc = (sum + (w >> 1)) >> l2w;
    else dc = (sum + (h >> 1)) >> l2h;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int v = dc;
            if (pdpc) {
                const int wt = pdpc_wgt(y, scale), wl = pdpc_wgt(x, scale);
                v = clip_bd((lft[1 + y] * wl + abv[1 + x] * wt + (64 - wl - wt) * v + 32) >> 6);
            }
            out[y * w + x] = (uint16_t)v;
        }
}

/* Vertical-class angular prediction of a w x h block: `main` = the arm along the prediction direction's source (above
 * for vertical modes), `side` the other one; both with index 0 = outermost corner sample of line `mrl`.
 * midx = distance of the mode from the pure direction (negative: the angle points to the side arm); the horizontal class
 * is this function on the transposed block.  is_luma selects the 4-tap fC / fG filters, chroma interpolates linearly.
 * smooth: the references were smoothed already (integer-slope modes with filterFlag); use_fg: Gaussian instead of cubic.
 * pdpc: PDPC allowed for this block (refIdx 0, not BDPCM, size rule) */
static void
pred_angular_v(uint16_t *main, const uint16_t *side, int w, int h, int l2w, int l2h, int midx, int is_luma, int use_fg, int mrl, int pdpc,
               int tl /* the corner sample pure-direction PDPC subtracts: always the ABOVE array's (rcn_intra_angular.c:308, :328) */,
               uint16_t *out, int ostride_x, int ostride_y)
{
    const int am = midx < 0 ? -midx : midx;
    const int angle = midx < 0 ? -ang_table[am] : ang_table[am];
    const int inv = inv_ang_table[am];
    (void)l2w;
    if (midx < 0) {
        /* extend the main arm to negative indices from the side arm: ref[x] = side[min((-x * inv + 256) >> 9, h)],
         * both arms indexed from the outermost corner sample (rcn_intra.c:120-125, :703-714) */
        int acc = 256;
        for (int k = -1; k >= -h; --k) {
            acc += inv;
            int sidx = acc >> 9;
            if (sidx > h) sidx = h;
            main[k] = side[sidx];
        }
    }
    const uint16_t *ref = main;                  /* ref[x] = p[-1 - refIdx + x][-1 - refIdx] */
    int nscale = -1;
    if (pdpc) {
        if (am == 0) nscale = (l2w + l2h - 2) >> 2;
        else if (midx > 0) nscale = 2 < (l2h - (ilog2(3 * inv - 2) - 8)) ? 2 : (l2h - (ilog2(3 * inv - 2) - 8));
    }
    for (int y = 0; y < h; ++y) {
        const int pos = (y + 1 + mrl) * angle;
        const int iidx = (pos >> 5) + mrl, ifact = pos & 31;
        for (int x = 0; x < w; ++x) {
            int v;
            if (is_luma) {
                if (!(angle & 31)) {
                    /* integer slope: plain copy (fC[0]; where filterFlag held, the references were smoothed instead) */
                    v = ref[x + iidx + 1];
                } else {
                    const int8_t *f = use_fg ? fg_taps[ifact] : fc_taps[ifact];
                    v = clip_bd((f[0] * ref[x + iidx] + f[1] * ref[x + iidx + 1] + f[2] * ref[x + iidx + 2] + f[3] * ref[x + iidx + 3] + 32) >> 6);
                }
            } else {
                v = ifact ? ((32 - ifact) * ref[x + iidx + 1] + ifact * ref[x + iidx + 2] + 16) >> 5 : ref[x + iidx + 1];
            }
            if (nscale >= 0) {
                if (am == 0) {
                    /* pure vertical: refL = p[-1][y] - p[-1][-1] + pred, wL = 32 >> ((x << 1) >> nScale) */
                    const int wl = pdpc_wgt(x, nscale);
                    v = clip_bd((((int)side[1 + y] - tl + v) * wl + (64 - wl) * v + 32) >> 6);
                } else if (x < (3 << nscale)) {
                    const int wl = pdpc_wgt(x, nscale);
                    const int dy = y + (((x + 1) * inv + 256) >> 9);
                    v = clip_bd((side[1 + dy] * wl + (64 - wl) * v + 32) >> 6);
                }
            }
            out[y * ostride_y + x * ostride_x] = (uint16_t)v;
        }
    }
}

/* wide-angle remap (8.4.5.2.6; derive_wide_angular_mode, rcn_intra.c:54-66).  Modes beyond 66 are the spec's 67 .. 80; below
 * 2 the reference's numbering is used (spec mode -k = 2 - k here: 1, 0, -1 .. -12), which keeps "distance from mode 18" equal
 * to the index of the angle tables */
static int wide_angle(int l2w, int l2h, int mode)
{
    if (l2w == l2h || mode < 2 || mode > 66) return mode;
    const int r = l2w > l2h ? l2w - l2h : l2h - l2w;
    if (l2w > l2h && mode < (r > 1 ? 8 + 2 * r : 8)) return mode + 65;
    if (l2h > l2w && mode > (r > 1 ? 60 - 2 * r : 60)) return mode - 65;
    return mode;
}

/* One luma / chroma angular, planar or DC prediction into out[h][w].  flags: see ovhip_itask. */
static void
pred_regular(const uint16_t *plane, int stride, const ovhip_itask *t, int is_luma, uint16_t *out)
{
    const int l2w = t->log2_w, l2h = t->log2_h, w = 1 << l2w, h = 1 << l2h;
    const int mrl = is_luma ? t->mrl_idx : 0;
    const int unit = is_luma ? 4 : 2;
    static _Thread_local ref_bufs R, F;       /* (bench.py runs one decode per host thread) */
    uint16_t *abv = R.a + REF_PAD, *lft = R.l + REF_PAD;
    fetch_refs(plane, stride, t->x, t->y, w, h, unit, !!(t->flags & OVHIP_IF_CORNER), t->avl_abv, t->avl_lft, mrl, abv, lft);
    const int bdpcm = !!(t->flags & OVHIP_IF_BDPCM);
    /* PDPC: refIdx 0, no BDPCM, and chroma blocks at least 4x4 (rcn_intra.c:1088, :1119; luma intra blocks always are) */
    const int pdpc_ok = !mrl && !bdpcm && (is_luma || (l2w > 1 && l2h > 1));
    if (bdpcm) {
        /* pure copy along the DPCM direction (rcn_intra.c:511-523) */
        const int vertical = !!(t->flags & OVHIP_IF_BDPCM_VER);
        for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) out[y * w + x] = vertical ? abv[1 + x] : lft[1 + y];
        return;
    }
    int mode = t->mode;
    if (mode == 0) {
        const uint16_t *a = abv + mrl, *l = lft + mrl;
        if (is_luma && !mrl && l2w + l2h > 5) {
            uint16_t *fa = F.a + REF_PAD, *fl = F.l + REF_PAD;
            smooth_refs(abv, lft, fa, w + 4); smooth_refs(lft, abv, fl, h + 4);
            a = fa; l = fl;
        }
        pred_planar(a, l, l2w, l2h, pdpc_ok, out);
        return;
    }
    if (mode == 1) { pred_dc(abv + mrl, lft + mrl, l2w, l2h, pdpc_ok, out); return; }
    mode = wide_angle(l2w, l2h, mode);
    const int vertical = mode >= 34;
    const int midx = vertical ? mode - 50 : 18 - mode;
    const int am = midx < 0 ? -midx : midx;
    const int angle = ang_table[am];
    int use_fg = 0;
    uint16_t *m = vertical ? abv : lft, *s = vertical ? lft : abv;
    if (is_luma && !mrl && l2w + l2h > 5 && am > hv_dist_thres[(l2w + l2h) >> 1]) {
        if (!(angle & 31)) {
            /* integer slope: [1 2 1] smoothed references, no interpolation (refFilterFlag) */
            uint16_t *fa = F.a + REF_PAD, *fl = F.l + REF_PAD;
            smooth_refs(abv, lft, fa, 2 * w); smooth_refs(lft, abv, fl, 2 * h);
            m = vertical ? fa : fl; s = vertical ? fl : fa;
        } else {
            use_fg = 1;
        }
    }
    if (vertical) pred_angular_v(m, s, w, h, l2w, l2h, midx, is_luma, use_fg, mrl, pdpc_ok, abv[0], out, 1, w);
    else          pred_angular_v(m, s, h, w, l2h, l2w, midx, is_luma, use_fg, mrl, pdpc_ok, abv[0], out, w, 1);
}

/* ---------------------------------------------------------------- intra sub-partitions (intra_pred_isp, rcn_intra.c:566-640)
 * One reference arm of an ISP prediction call, as fill_ref_above_0 / fill_ref_left_0 (rcn_fill_ref.c:71-150, :320-390) build it
 * for the CODING UNIT's geometry and intra_pred_isp then shifts it by the partition's offset:
 *   S(k)  = the sample k steps along the CU's arm from its corner (above: (cu_x - 1 + k, y - 1); left: (x - 1, cu_y - 1 + k))
 *   c     = the corner unit is available in this arm's own progress map, a = units available behind it (highest set bit)
 *   other = the other arm's map has any bit set, F = first sample of the other arm at this partition ((x - 1, y) / (x, y - 1))
 *   out[k] = value(k + off) for k <= cb + pb, replicated beyond (rcn_intra.c:597-603) */
static void
isp_arm(const uint16_t *p0, int step, int cb, int pb, int off, int c, int a, int other, int F, uint16_t *out)
{
    const int n_cu = 2 * cb + 1, nb_ref = (2 * cb) / 4 + 1, len = cb + pb;
    uint16_t cu[2 * 64 + 1 + 8];
#define S(k) p0[(k) * step]
    if (c && a >= nb_ref) {
        for (int k = 0; k < n_cu; ++k) cu[k] = S(k);
    } else if (c || a) {
        int last = 4 * a < n_cu + 3 ? 4 * a : n_cu + 3;
        cu[0] = c ? S(0) : S(1);
        for (int k = 1; k <= last; ++k) cu[k] = S(k);
        for (int k = last + 1; k < n_cu + 4; ++k) cu[k] = cu[last];
    } else {
        for (int k = 0; k < n_cu + 4; ++k) cu[k] = (uint16_t)(other ? F : 1 << (BD - 1));
    }
#undef S
    for (int k = n_cu; k < n_cu + 8; ++k) cu[k] = cu[n_cu - 1];                 /* "padding for wide angle" */
    for (int k = 0; k <= len; ++k) out[k] = cu[k + off];
    for (int k = len + 1; k < len + 1 + REF_PAD; ++k) out[k] = out[len];
}

static void
pred_isp(const uint16_t *plane, int stride, const ovhip_itask *t, uint16_t *out)
{
    const int l2w = t->log2_w, l2h = t->log2_h, w = 1 << l2w, h = 1 << l2h;
    const int cbw = 1 << t->isp_log2_cb_w, cbh = 1 << t->isp_log2_cb_h, ox = t->isp_off_x, oy = t->isp_off_y;
    const int ca = !!(t->flags & OVHIP_IF_CORNER), cl = !!(t->flags & OVHIP_IF_CORNER_L);
    static _Thread_local ref_bufs R;
    uint16_t *abv = R.a + REF_PAD, *lft = R.l + REF_PAD;
    const uint16_t *here = plane + t->y * stride + t->x;
    /* (the other arm's first sample is used only when that arm has anything: not read at the picture's left column / top row, where
     *  it would lie in front of the plane) */
    isp_arm(here - stride - ox - 1, 1, cbw, w, ox, ca, t->avl_abv, cl || t->avl_lft, t->x ? here[-1] : 0, abv);
    isp_arm(here - (oy + 1) * stride - 1, stride, cbh, h, oy, cl, t->avl_lft, ca || t->avl_abv, t->y ? here[-stride] : 0, lft);
    const int pdpc_ok = l2h > 1;                       /* rcn_intra.c:608, :618; cubic_v / cubic_h: log2_pb_h > 1 */
    int mode = t->mode;
    if (mode == 0) { pred_planar(abv, lft, l2w, l2h, pdpc_ok, out); return; }
    if (mode == 1) { pred_dc(abv, lft, l2w, l2h, pdpc_ok, out); return; }
    mode = wide_angle(t->isp_log2_cb_w, t->isp_log2_cb_h, mode);         /* by the CU's shape (:627) */
    const int vertical = mode >= 34;
    const int midx = vertical ? mode - 50 : 18 - mode;
    /* always the cubic filter, never smoothed references (intra_angular_cubic_v / _h) */
    if (vertical) pred_angular_v(abv, lft, w, h, l2w, l2h, midx, 1, 0, 0, pdpc_ok, abv[0], out, 1, w);
    else          pred_angular_v(lft, abv, h, w, l2h, l2w, midx, 1, 0, 0, pdpc_ok, abv[0], out, w, 1);
}

/* ---------------------------------------------------------------- matrix-based intra prediction (rcn_intra_mip.c:44-400) */
static void
mip_upsample(uint16_t *dst, const uint16_t *src, const uint16_t *ref, int l2_up_src, int l2_opp, int src_step, int src_stride,
             int dst_step, int dst_stride, int ref_step, int l2_scale)
{
    const int rnd = 1 << (l2_scale - 1);
    const uint16_t *src_line = src, *bnd = ref + ref_step;
    uint16_t *dst_line = dst;
    for (int i = 0; i < (1 << l2_opp); ++i) {
        const uint16_t *before = bnd, *after = src_line;
        uint16_t *d = dst_line;
        for (int j = 0; j < (1 << l2_up_src); ++j) {
            int32_t bv = (*before) << l2_scale, av = 0;
            for (int pos = 1; pos <= (1 << l2_scale); ++pos) {
                bv -= *before; av += *after;
                *d = (uint16_t)((bv + av + rnd) >> l2_scale);
                d += dst_step;
            }
            before = after; after += src_step;
        }
        src_line += src_stride; dst_line += dst_stride; bnd += ref_step;
    }
}

static void
pred_mip(const uint16_t *plane, int stride, const ovhip_itask *t, uint16_t *out)
{
    const int l2w = t->log2_w, l2h = t->log2_h, w = 1 << l2w;
    const int tr = !!(t->flags & OVHIP_IF_MIP_TR), mode = t->mode;
    static _Thread_local ref_bufs R;
    uint16_t *abv = R.a + REF_PAD, *lft = R.l + REF_PAD;
    fetch_refs(plane, stride, t->x, t->y, w, 1 << l2h, 4, !!(t->flags & OVHIP_IF_CORNER), t->avl_abv, t->avl_lft, 0, abv, lft);
    const int l2b = 1 << ((l2w > 2) || (l2h > 2));               /* log2 of the reduced boundary length per side: 1 or 2 */
    const int nb = 1 << l2b, l2bx = l2w - l2b, l2by = l2h - l2b;
    const int red = l2h == 2 || l2w == 2 || (l2h <= 3 && l2w <= 3);
    int16_t bnd[8];
    for (int j = 0; j < nb; ++j) {
        int sa = 0, sl = 0;
        for (int i = 0; i < (1 << l2bx); ++i) sa += abv[1 + i + (j << l2bx)];
        for (int i = 0; i < (1 << l2by); ++i) sl += lft[1 + i + (j << l2by)];
        const int va = (sa + ((1 << l2bx) >> 1)) >> l2bx, vl = (sl + ((1 << l2by) >> 1)) >> l2by;
        if (!tr) { bnd[j] = (int16_t)va; bnd[nb + j] = (int16_t)vl; }
        else     { bnd[nb + j] = (int16_t)va; bnd[j] = (int16_t)vl; }
    }
    const int in_off = bnd[0];
    if (red) bnd[0] = 1 << (BD - 1);
    int sum = 0;
    for (int i = 0; i < 2 * nb; ++i) { bnd[i] = (int16_t)(bnd[i] - in_off); sum += bnd[i]; }
    const int rnd_mip = 32 - 32 * sum;
    const int l2rw = red ? 2 : (l2w < 3 ? l2w : 3), l2rh = red ? 2 : (l2h < 3 ? l2h : 3);
    const uint8_t *mat;
    if (l2w == 2 && l2h == 2) mat = ovt_mip_4x4 + mode * 16 * 4;
    else if (red)             mat = ovt_mip_8x8 + mode * 16 * 8;
    else                      mat = ovt_mip_16x16 + mode * 64 * 8;
    uint16_t mp[64], mp2[64];
    const int sx = 2 * nb;
    for (int i = 0, pos = 0; i < (1 << (l2rw + l2rh)); ++i, mat += sx) {
        int v = 0;
        for (int k = 0; k < sx; ++k) v += bnd[k] * mat[k];
        mp[pos++] = (uint16_t)clip_bd(((v + rnd_mip) >> 6) + in_off);
    }
    const uint16_t *red_pred = mp;
    if (tr) {
        for (int i = 0; i < (1 << l2rh); ++i) for (int j = 0; j < (1 << l2rw); ++j) mp2[j + (i << l2rw)] = mp[(j << l2rh) + i];
        red_pred = mp2;
    }
    const int sxs = l2w - l2rw, sys = l2h - l2rh;
    if (!sxs && !sys) {
        for (int i = 0; i < (1 << l2rh); ++i) for (int j = 0; j < (1 << l2rw); ++j) out[i * w + j] = red_pred[(i << l2rw) + j];
        return;
    }
    const uint16_t *src; int src_step, src_stride;
    if (sxs) {
        uint16_t *d0 = out + ((1 << sys) - 1) * w;
        mip_upsample(d0, red_pred, lft, l2rw, l2rh, 1, 1 << l2rw, 1, (1 << sys) * w, 1 << sys, sxs);
        src = d0; src_step = (1 << sys) * w; src_stride = 1;
    } else {
        src = red_pred; src_step = w; src_stride = 1;
    }
    if (sys) mip_upsample(out, src, abv, l2rh, l2w, src_step, src_stride, w, 1, 1, sys);
}

/* ---------------------------------------------------------------- cross-component linear model (rcn_intra_cclm.c:56-880) */
typedef struct { int a, b, shift; } lm_par;

static lm_par lm_params(int min_l, int min_c, int max_c, int v, int l2rng)
{
    lm_par p;
    const int rc = max_c - min_c;
    const int l2c = rc ? ilog2(rc < 0 ? -rc : rc) + 1 : 0;
    int a = (rc * v + ((1 << l2c) >> 1)) >> l2c;
    int shift = 3 + l2rng - l2c;
    if (shift < 1) { shift = 1; a = a ? (a < 0 ? -15 : 15) : 0; }
    p.a = a; p.shift = shift; p.b = min_c - ((a * min_l) >> shift);
    return p;
}

static void
pred_cclm(const oracle_pic *pic, const ovhip_itask *t, int log2_ctu, uint16_t *out_cb, uint16_t *out_cr)
{
    const int l2w = t->log2_w, l2h = t->log2_h, w = 1 << l2w, h = 1 << l2h, x0 = t->x, y0 = t->y;
    const int sl = pic->stride_y, sc = pic->stride_c;
    const uint16_t *sy = pic->y + (y0 * 2) * sl + x0 * 2, *scb = pic->cb + y0 * sc + x0, *scr = pic->cr + y0 * sc + x0;
    const int mode = t->mode;                                       /* 67 LM, 68 MDLM left, 69 MDLM top */
    const int lft_avail = mode == 68 ? t->avl_lft > 0 : !!t->avl_lft, abv_avail = mode == 69 ? t->avl_abv > 0 : !!t->avl_abv;
    const int first_line = !((y0 * 2) & ((1 << log2_ctu) - 1));    /* top row of the CTU: one luma line above only */
    uint16_t py[4], pcb[4], pcr[4];
    int n = 0;
    lm_par pb = { 0, 1 << (BD - 1), 0 }, pr = { 0, 1 << (BD - 1), 0 };
    int n_abv = 0, abv_step = 1, n_lft = 0, lft_step = 1;
    if (mode == 67 && (abv_avail || lft_avail)) {
        if (abv_avail) { const int l2n = abv_avail + !lft_avail; abv_step = (w >> l2n) > 1 ? (w >> l2n) : 1; n_abv = (abv_avail + !lft_avail) << 1; if (n_abv > w) n_abv = w; }
        if (lft_avail) { const int l2n = lft_avail + !abv_avail; lft_step = (h >> l2n) > 1 ? (h >> l2n) : 1; n_lft = (lft_avail + !abv_avail) << 1; if (n_lft > h) n_lft = h; }
    } else if (mode == 69 && abv_avail) {
        const int len = t->avl_abv << 1;                            /* contiguous available reference length above */
        n_abv = len < 4 ? len : 4; abv_step = (len >> 2) > 1 ? (len >> 2) : 1;
    } else if (mode == 68 && lft_avail) {
        const int len = t->avl_lft << 1;
        n_lft = len < 4 ? len : 4; lft_step = (len >> 2) > 1 ? (len >> 2) : 1;
    }
    {
        int pad_left;
        const int sp = abv_step >> 1;
        const uint16_t *s = first_line ? sy - sl + (sp << 1) : sy - 2 * sl + (sp << 1);
        pad_left = sp == 0 && !lft_avail;
        for (int i = 0; i < n_abv; ++i) {
            int v;
            if (first_line) v = (2 + s[-(!pad_left)] + 2 * s[0] + s[1]) >> 2;
            else v = (4 + s[-(!pad_left)] + 2 * s[0] + s[1] + s[sl - (!pad_left)] + 2 * s[sl] + s[sl + 1]) >> 3;
            py[n] = (uint16_t)v; pcb[n] = scb[-sc + sp + i * abv_step]; pcr[n] = scr[-sc + sp + i * abv_step]; ++n;
            s += abv_step << 1; pad_left = 0;
        }
    }
    {
        const int sp = lft_step >> 1;
        const uint16_t *s = sy - 2 + sp * 2 * sl;
        for (int i = 0; i < n_lft; ++i) {
            const int v = (4 + 2 * s[0] + s[1] + s[-1] + 2 * s[sl] + s[sl + 1] + s[sl - 1]) >> 3;
            py[n] = (uint16_t)v; pcb[n] = scb[-1 + (sp + i * lft_step) * sc]; pcr[n] = scr[-1 + (sp + i * lft_step) * sc]; ++n;
            s += 2 * sl * lft_step;
        }
    }
    if (n) {
        int min_l, max_l, min_cb, max_cb, min_cr, max_cr;
        if (n == 2) {
            const int mi = py[0] >= py[1], ma = !mi;
            min_l = py[mi]; max_l = py[ma]; min_cb = pcb[mi]; max_cb = pcb[ma]; min_cr = pcr[mi]; max_cr = pcr[ma];
        } else {
            int idx[4] = { 0, 2, 1, 3 }, *mn = &idx[0], *mx = &idx[2], tswap, *pswap;
            if (py[0] > py[2]) { tswap = mn[0]; mn[0] = mn[1]; mn[1] = tswap; }
            if (py[1] > py[3]) { tswap = mx[0]; mx[0] = mx[1]; mx[1] = tswap; }
            if (py[mn[0]] > py[mx[1]]) { pswap = mn; mn = mx; mx = pswap; }
            if (py[mn[1]] > py[mx[0]]) { tswap = mn[1]; mn[1] = mx[0]; mx[0] = tswap; }
            min_l = (py[mn[0]] + py[mn[1]] + 1) >> 1; max_l = (py[mx[0]] + py[mx[1]] + 1) >> 1;
            min_cb = (pcb[mn[0]] + pcb[mn[1]] + 1) >> 1; max_cb = (pcb[mx[0]] + pcb[mx[1]] + 1) >> 1;
            min_cr = (pcr[mn[0]] + pcr[mn[1]] + 1) >> 1; max_cr = (pcr[mx[0]] + pcr[mx[1]] + 1) >> 1;
        }
        pb.a = 0; pb.b = min_cb; pb.shift = 0; pr.a = 0; pr.b = min_cr; pr.shift = 0;
        const int rl = max_l - min_l;
        if (rl) {
            static const uint8_t div_lut[16] = { 0, 7, 6, 5, 5, 4, 4, 3, 3, 2, 2, 1, 1, 1, 1, 0 };
            int l2r = ilog2(rl);
            const int nd = ((rl << 4) >> l2r) & 15, v = div_lut[nd] | 8;
            l2r += nd != 0;
            pb = lm_params(min_l, min_cb, max_cb, v, l2r);
            pr = lm_params(min_l, min_cr, max_cr, v, l2r);
        }
    }
    for (int j = 0; j < h; ++j)
        for (int i = 0; i < w; ++i) {
            const uint16_t *s = sy + 2 * j * sl + 2 * i;
            const int pl = i == 0 && !lft_avail;
            const int v = (4 + s[1] + s[-(!pl)] + 2 * s[0] + 2 * s[sl] + s[sl + 1] + s[sl - (!pl)]) >> 3;
            out_cb[j * w + i] = (uint16_t)clip_bd(((v * pb.a) >> pb.shift) + pb.b);
            out_cr[j * w + i] = (uint16_t)clip_bd(((v * pr.a) >> pr.shift) + pr.b);
        }
}

/* ---------------------------------------------------------------- ordered pass */
static int res_scale(int v, int scale)
{
    const int sign = v & (1 << 15);
    int a = (clip_bd(abs(v)) * scale + (1 << 10)) >> 11;
    return clip3i(sign ? -a : a, -(1 << 15), 1 << 15);
}

/* ciip_wt != 0: the block holds an inter prediction that the intra (planar) one is blended into first
 * (rcn_ciip_weighted_sum, rcn_inter.c:2968-3009; put_weighted_ciip_pixels rcn_mc.c:1611-1628) */
static void
store_block(uint16_t *dst, int dstride, const uint16_t *pred, const int16_t *res, int rstride, int w, int h, int scaled, int scale, int ciip_wt)
{
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int v = pred[y * w + x];
            if (ciip_wt) v = (v * ciip_wt + dst[y * dstride + x] * (4 - ciip_wt) + 2) >> 2;
            if (res) { int r = res[y * rstride + x]; if (scaled) r = res_scale(r, scale); v = clip_bd(v + r); }
            dst[y * dstride + x] = (uint16_t)v;
        }
}


/* Executes the ordered tasks in array order (the recorder emits them in decoding order = a valid topological order; the
 * device runs them level by level).  res: the residuals the transform stage STOREd for these blocks; scales: in/out, the
 * device-derived chroma residual scales (regions of ordered tasks are filled in here). */
void
oracle_intra_tasks(const oracle_pic *pic, const oracle_res *res, const ovhip_itask *tasks, uint32_t n, const ovhip_lmcs_region *regs,
                   const ovhip_lmcs_luts *luts, int16_t *scales, int log2_ctu)
{
    static _Thread_local uint16_t pa[128 * 128], pb[128 * 128];
    for (uint32_t i = 0; i < n; ++i) {
        const ovhip_itask *t = &tasks[i];
        const int w = 1 << t->log2_w, h = 1 << t->log2_h;
        const int scaled = !!(t->flags & OVHIP_IF_RES_SCALE);
        const int scale = scaled ? ((t->flags & OVHIP_IF_SCALE_IDX) ? scales[t->c_scale] : t->c_scale) : 0;
        switch (t->kind) {
        case OVHIP_IT_LUMA: {
            if (t->flags & OVHIP_IF_ISP) pred_isp(pic->y, pic->stride_y, t, pa);
            else if (t->flags & OVHIP_IF_MIP) pred_mip(pic->y, pic->stride_y, t, pa);
            else pred_regular(pic->y, pic->stride_y, t, 1, pa);
            if ((t->flags & OVHIP_IF_ISP) && (t->flags & OVHIP_IF_RES_Y)) {
                /* only the partitions that carry a residual add one (the residual picture holds nothing defined elsewhere) */
                const int pbw = 1 << t->isp_log2_pb;
                for (int x = 0; x < w; x += pbw) {
                    const int on = (t->isp_res_mask >> (x >> t->isp_log2_pb)) & 1;
                    uint16_t col[64 * 64];
                    for (int y = 0; y < h; ++y) for (int q = 0; q < pbw; ++q) col[y * pbw + q] = pa[y * w + x + q];
                    store_block(pic->y + t->y * pic->stride_y + t->x + x, pic->stride_y, col,
                                on ? res->y + t->y * res->stride_y + t->x + x : NULL, res->stride_y, pbw, h, 0, 0, 0);
                }
                break;
            }
            store_block(pic->y + t->y * pic->stride_y + t->x, pic->stride_y, pa,
                        (t->flags & OVHIP_IF_RES_Y) ? res->y + t->y * res->stride_y + t->x : NULL, res ? res->stride_y : 0, w, h, 0, 0, t->ciip_wt);
            break;
        }
        case OVHIP_IT_CHROMA: {
            if (t->mode >= 67) pred_cclm(pic, t, log2_ctu, pa, pb);
            else { pred_regular(pic->cb, pic->stride_c, t, 0, pa); pred_regular(pic->cr, pic->stride_c, t, 0, pb); }
            store_block(pic->cb + t->y * pic->stride_c + t->x, pic->stride_c, pa,
                        (t->flags & OVHIP_IF_RES_CB) ? res->cb + t->y * res->stride_c + t->x : NULL, res ? res->stride_c : 0, w, h, scaled, scale, t->ciip_wt);
            store_block(pic->cr + t->y * pic->stride_c + t->x, pic->stride_c, pb,
                        (t->flags & OVHIP_IF_RES_CR) ? res->cr + t->y * res->stride_c + t->x : NULL, res ? res->stride_c : 0, w, h, scaled, scale, t->ciip_wt);
            break;
        }
        case OVHIP_IT_REGION:
            lmcs_scale_regions(pic, &regs[t->c_scale], 1, luts, &scales[t->c_scale], 1);
            break;
        case OVHIP_IT_RES_C: {
            /* chroma residual of a block predicted earlier (inter), whose scale only became known in the ordered pass */
            for (int p = 1; p < 3; ++p) {
                if (!(t->flags & (p == 1 ? OVHIP_IF_RES_CB : OVHIP_IF_RES_CR))) continue;
                uint16_t *d = (p == 1 ? pic->cb : pic->cr) + t->y * pic->stride_c + t->x;
                const int16_t *r = (p == 1 ? res->cb : res->cr) + t->y * res->stride_c + t->x;
                for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {
                    int v = r[y * res->stride_c + x];
                    if (scaled) v = res_scale(v, scale);
                    d[y * pic->stride_c + x] = (uint16_t)clip_bd(d[y * pic->stride_c + x] + v);
                }
            }
            break;
        }
        default: break;
        }
    }
}
