// kernels_intra.hip -- the ordered pass on gfx950: intra prediction (+ CIIP's planar part, ordered chroma-scale regions and the
// chroma residuals that hang on them), one LAUNCH PER LEVEL (ovhip_itask.level), one wavefront per task.
//
// Replaces, for every block of an intra CU, the reference's per-block chain
//   rcn_intra_tu / rcn_tu_st / rcn_tu_c (rcn_transform_tree.c:1384-1430, :1269-1287, :1349-1382)
//     -> intra_pred / intra_pred_mrl / mip.rcn_intra_mip / intra_pred_c (-> cclm.*)   (rcn_intra.c:484-1180, rcn_intra_mip.c,
//        rcn_intra_cclm.c, rcn_fill_ref.c, rcn_intra_angular.c, rcn_intra_dc_planar.c)
//     -> ict.add / ict.ict of the block's residual                                     (rcn_residuals.c:46-222)
// The reference gets the order for free (decoding order on one core); here the recorder computed for every task the level
// of its inputs, a level's tasks are mutually independent, and the launch boundary between levels is what makes one
// level's stores visible to the next (per-XCD L2s are not coherent inside a launch).
//
// Mapping: lane = sample (w * h / 64 samples per lane).  Reference samples of the block are fetched once into LDS with the
// substitution rules of 8.4.5.2.8 (availability = unit counts from the recorder), smoothed copies next to them when the mode
// asks for them; every prediction mode then reads LDS only.  int16 / int32 arithmetic, no MFMA: per-sample stencils.
#include "ovvc_common.hip.h"
#define OVT_ATTR __device__
#include "vvc_mip_tables.h"

namespace {

__device__ const short g_ang[32] = { 0, 1, 2, 3, 4, 6, 8, 10, 12, 14, 16, 18, 20, 23, 26, 29, 32, 35, 39, 45, 51, 57, 64, 73, 86, 102, 128, 171, 256, 341, 512, 1024 };
__device__ const short g_inv_ang[32] = { 0, 16384, 8192, 5461, 4096, 2731, 2048, 1638, 1365, 1170, 1024, 910, 819, 712, 630, 565, 512, 468, 420, 364,
                                         321, 287, 256, 224, 191, 161, 128, 96, 64, 48, 32, 16 };
__device__ const unsigned char g_hv_thres[8] = { 24, 24, 24, 14, 2, 0, 0, 0 };
__device__ const signed char g_fc[32][4] = {
    { 0, 64, 0, 0 }, { -1, 63, 2, 0 }, { -2, 62, 4, 0 }, { -2, 60, 7, -1 }, { -2, 58, 10, -2 }, { -3, 57, 12, -2 }, { -4, 56, 14, -2 },
    { -4, 55, 15, -2 }, { -4, 54, 16, -2 }, { -5, 53, 18, -2 }, { -6, 52, 20, -2 }, { -6, 49, 24, -3 }, { -6, 46, 28, -4 }, { -5, 44, 29, -4 },
    { -4, 42, 30, -4 }, { -4, 39, 33, -4 }, { -4, 36, 36, -4 }, { -4, 33, 39, -4 }, { -4, 30, 42, -4 }, { -4, 29, 44, -5 }, { -4, 28, 46, -6 },
    { -3, 24, 49, -6 }, { -2, 20, 52, -6 }, { -2, 18, 53, -5 }, { -2, 16, 54, -4 }, { -2, 15, 55, -4 }, { -2, 14, 56, -4 }, { -2, 12, 57, -3 },
    { -2, 10, 58, -2 }, { -1, 7, 60, -2 }, { 0, 4, 62, -2 }, { 0, 2, 63, -1 } };

#define IR_NEG  64                       // room for the negative-angle extension of the main arm
#define IR_LEN  (IR_NEG + 2 * 64 + 4 + 24)   // + reference line offset + wide-angle tail
struct IntraLds {
    uint16_t abv[IR_LEN], lft[IR_LEN];   // index IR_NEG = outermost corner sample of the reference line
    uint16_t fabv[IR_LEN], flft[IR_LEN]; // [1 2 1]-smoothed copies
    uint16_t red[64], hup[8 * 64];       // MIP: reduced prediction, horizontally up-sampled rows
    int      par[16];                    // CCLM parameters / MIP boundary
};

__device__ __forceinline__ int pdpc_wgt(int i, int scale) { const int sh = (i << 1) >> scale; return sh > 5 ? 0 : 32 >> sh; }
__device__ __forceinline__ int ilog2(int v) { return 31 - __clz(v); }

// reference samples with substitution: see fetch_refs in oracle/ovvc_oracle_intra.c for the sequential form; with the availability
// the recorder hands over (corner flag + unit counts from the corner outwards) every sample's source is known in closed form
__device__ void fetch_refs(IntraLds &s, const uint16_t *__restrict__ plane, int stride, int x0, int y0, int w, int h, int unit, bool corner,
                           int avl_abv, int avl_lft, int mrl, int lane)
{
    const int na = 2 * w + mrl + 1, nl = 2 * h + mrl + 1;
    const uint16_t *org = plane + (y0 - 1 - mrl) * stride + (x0 - 1 - mrl);       // sample (k = 0) of both arms
    // fall-back values (wave-uniform): first sample of each arm's block part, bottom-most corner sample
    const int a1 = avl_abv ? org[mrl + 1] : 0, l1 = avl_lft ? org[(mrl + 1) * stride] : 0;
    const int none = !corner && !avl_abv && !avl_lft;
    const int la = min(mrl + avl_abv * unit, na - 1), ll = min(mrl + avl_lft * unit, nl - 1);   // last available sample per arm
    for (int k = lane; k < na + 24; k += 64) {
        int v;
        const int kk = min(k, na - 1);
        if (none) v = 1 << (OV_BD - 1);
        else if (kk <= mrl) v = corner ? org[kk] : ((mrl == 0 && avl_abv && avl_lft) ? a1 : (avl_lft ? l1 : a1));
        else if ((kk - mrl - 1) / unit < avl_abv) v = org[kk];
        else v = avl_abv ? org[la] : (corner ? org[mrl] : l1);
        s.abv[IR_NEG + k] = (uint16_t)v;
    }
    for (int k = lane; k < nl + 24; k += 64) {
        int v;
        const int kk = min(k, nl - 1);
        if (none) v = 1 << (OV_BD - 1);
        else if (kk <= mrl) v = corner ? org[kk * stride] : (avl_lft ? l1 : a1);
        else if ((kk - mrl - 1) / unit < avl_lft) v = org[kk * stride];
        else v = avl_lft ? org[ll * stride] : (corner ? org[mrl * stride] : a1);
        s.lft[IR_NEG + k] = (uint16_t)v;
    }
}

// filter_ref_samples (rcn_fill_ref.c:41-68) on both arms: [0] from the two arms, [1 2 1] up to len - 1, the rest copied
__device__ void smooth_refs(IntraLds &s, int len_a, int len_l, int lane)
{
    const uint16_t *a = s.abv + IR_NEG, *l = s.lft + IR_NEG;
    for (int i = lane; i < 2 * 64 + 4 + 24; i += 64) {
        int va, vl;
        if (i == 0) { va = (l[1] + 2 * a[0] + a[1] + 2) >> 2; vl = (a[1] + 2 * l[0] + l[1] + 2) >> 2; }
        else {
            va = i < len_a ? (a[i + 1] + 2 * a[i] + a[i - 1] + 2) >> 2 : a[i];
            vl = i < len_l ? (l[i + 1] + 2 * l[i] + l[i - 1] + 2) >> 2 : l[i];
        }
        s.fabv[IR_NEG + i] = (uint16_t)va; s.flft[IR_NEG + i] = (uint16_t)vl;
    }
}

struct Sink {                 // where a predicted sample goes: blend (CIIP), residual, clip, store
    uint16_t *dst; int dstride;
    const int16_t *res; int rstride;     // nullptr: no residual
    int scaled, scale, ciip_wt;
};

__device__ __forceinline__ int res_scale(int v, int scale)
{
    const int sign = v & (1 << 15);
    int a = (ov_clip_bd(abs(v)) * scale + (1 << 10)) >> 11;
    return ov_clip3(sign ? -a : a, -(1 << 15), 1 << 15);
}

__device__ __forceinline__ void emit(const Sink &k, int x, int y, int v)
{
    uint16_t *d = k.dst + y * k.dstride + x;
    if (k.ciip_wt) v = (v * k.ciip_wt + (int)*d * (4 - k.ciip_wt) + 2) >> 2;
    if (k.res) {
        int r = k.res[y * k.rstride + x];
        if (k.scaled) r = res_scale(r, k.scale);
        v = ov_clip_bd(v + r);
    }
    *d = (uint16_t)v;
}

// planar / DC / angular / BDPCM prediction of one plane's block out of the LDS references
__device__ void pred_regular(IntraLds &s, const ovhip_itask &t, bool is_luma, const Sink &sink, int lane)
{
    const int l2w = t.log2_w, l2h = t.log2_h, w = 1 << l2w, h = 1 << l2h, n = w * h;
    const int mrl = is_luma ? t.mrl_idx : 0;
    const bool bdpcm = t.flags & OVHIP_IF_BDPCM;
    const bool pdpc_ok = !mrl && !bdpcm && (is_luma || (l2w > 1 && l2h > 1));
    const uint16_t *abv = s.abv + IR_NEG, *lft = s.lft + IR_NEG;
    if (bdpcm) {
        const bool ver = t.flags & OVHIP_IF_BDPCM_VER;
        for (int p = lane; p < n; p += 64) { const int x = p & (w - 1), y = p >> l2w; emit(sink, x, y, ver ? abv[1 + x] : lft[1 + y]); }
        return;
    }
    int mode = t.mode;
    if (mode == 0) {
        const uint16_t *a = abv + mrl, *l = lft + mrl;
        if (is_luma && !mrl && l2w + l2h > 5) { smooth_refs(s, w + 4, h + 4, lane); __syncthreads(); a = s.fabv + IR_NEG; l = s.flft + IR_NEG; }
        const int scale = (l2w + l2h - 2) >> 2;
        for (int p = lane; p < n; p += 64) {
            const int x = p & (w - 1), y = p >> l2w;
            const int pv = ((h - 1 - y) * a[1 + x] + (y + 1) * l[1 + h]) << l2w;
            const int ph = ((w - 1 - x) * l[1 + y] + (x + 1) * a[1 + w]) << l2h;
            int v = (pv + ph + n) >> (l2w + l2h + 1);
            if (pdpc_ok) { const int wt = pdpc_wgt(y, scale), wl = pdpc_wgt(x, scale); v = ov_clip_bd((l[1 + y] * wl + a[1 + x] * wt + (64 - wl - wt) * v + 32) >> 6); }
            emit(sink, x, y, v);
        }
        return;
    }
    if (mode == 1) {
        const uint16_t *a = abv + mrl, *l = lft + mrl;
        int sum = 0;
        if (w >= h) for (int x = lane; x < w; x += 64) sum += a[1 + x];
        if (h >= w) for (int y = lane; y < h; y += 64) sum += l[1 + y];
#pragma unroll
        for (int m = 32; m; m >>= 1) sum += __shfl_xor(sum, m);
        const int dc = w == h ? (sum + w) >> (l2w + 1) : (w > h ? (sum + (w >> 1)) >> l2w : (sum + (h >> 1)) >> l2h);
        const int scale = (l2w + l2h - 2) >> 2;
        for (int p = lane; p < n; p += 64) {
            const int x = p & (w - 1), y = p >> l2w;
            int v = dc;
            if (pdpc_ok) { const int wt = pdpc_wgt(y, scale), wl = pdpc_wgt(x, scale); v = ov_clip_bd((l[1 + y] * wl + a[1 + x] * wt + (64 - wl - wt) * v + 32) >> 6); }
            emit(sink, x, y, v);
        }
        return;
    }
    // wide-angle remap (derive_wide_angular_mode, rcn_intra.c:54-66; the reference's numbering below mode 2)
    if (l2w != l2h) {
        const int r = abs(l2w - l2h);
        if (l2w > l2h && mode < (r > 1 ? 8 + 2 * r : 8)) mode += 65;
        else if (l2h > l2w && mode > (r > 1 ? 60 - 2 * r : 60)) mode -= 65;
    }
    const bool vertical = mode >= 34;
    const int midx = vertical ? mode - 50 : 18 - mode, am = abs(midx);
    const int angle_abs = g_ang[am], inv = g_inv_ang[am];
    const int angle = midx < 0 ? -angle_abs : angle_abs;
    bool use_fg = false, smoothed = false;
    if (is_luma && !mrl && l2w + l2h > 5 && am > g_hv_thres[(l2w + l2h) >> 1]) {
        if (!(angle_abs & 31)) { smooth_refs(s, 2 * w, 2 * h, lane); __syncthreads(); smoothed = true; }
        else use_fg = true;
    }
    uint16_t *mainr = (vertical ? (smoothed ? s.fabv : s.abv) : (smoothed ? s.flft : s.lft)) + IR_NEG;
    const uint16_t *side = (vertical ? (smoothed ? s.flft : s.lft) : (smoothed ? s.fabv : s.abv)) + IR_NEG;
    const int mw = vertical ? w : h, mh = vertical ? h : w, l2mh = vertical ? l2h : l2w;     // block in the mode's own orientation
    if (midx < 0) {
        for (int k = 1 + lane; k <= mh; k += 64) { int si = (256 + k * inv) >> 9; si = min(si, mh); mainr[-k] = side[si]; }
        __syncthreads();
    }
    int nscale = -1;
    if (pdpc_ok) {
        if (am == 0) nscale = (l2w + l2h - 2) >> 2;
        else if (midx > 0) nscale = min(2, l2mh - (ilog2(3 * inv - 2) - 8));
    }
    const int tl = s.abv[IR_NEG];                         // pure-direction PDPC subtracts the ABOVE array's corner (rcn_intra_angular.c:308, :328)
    const int l2mw = vertical ? l2w : l2h;
    for (int p = lane; p < n; p += 64) {
        const int mx = p & (mw - 1), my = p >> l2mw;     // position in the mode's orientation
        const int pos = (my + 1 + mrl) * angle;
        const int iidx = (pos >> 5) + mrl, ifact = pos & 31;
        const uint16_t *r = mainr + mx + iidx;
        int v;
        if (is_luma) {
            if (!(angle_abs & 31)) v = r[1];
            else {
                int f0, f1, f2, f3;
                if (use_fg) { f0 = 16 - (ifact >> 1); f1 = 32 - (ifact >> 1); f2 = 16 + (ifact >> 1); f3 = ifact >> 1; }
                else { f0 = g_fc[ifact][0]; f1 = g_fc[ifact][1]; f2 = g_fc[ifact][2]; f3 = g_fc[ifact][3]; }
                v = ov_clip_bd((f0 * r[0] + f1 * r[1] + f2 * r[2] + f3 * r[3] + 32) >> 6);
            }
        } else {
            v = ifact ? ((32 - ifact) * r[1] + ifact * r[2] + 16) >> 5 : r[1];
        }
        if (nscale >= 0) {
            if (am == 0) { const int wl = pdpc_wgt(mx, nscale); v = ov_clip_bd((((int)side[1 + my] - tl + v) * wl + (64 - wl) * v + 32) >> 6); }
            else if (mx < (3 << nscale)) {
                const int wl = pdpc_wgt(mx, nscale), dy = my + (((mx + 1) * inv + 256) >> 9);
                v = ov_clip_bd((side[1 + dy] * wl + (64 - wl) * v + 32) >> 6);
            }
        }
        emit(sink, vertical ? mx : my, vertical ? my : mx, v);
    }
}

// matrix-based intra prediction (rcn_intra_mip.c:44-400)
__device__ void pred_mip(IntraLds &s, const ovhip_itask &t, const Sink &sink, int lane)
{
    const int l2w = t.log2_w, l2h = t.log2_h, w = 1 << l2w, h = 1 << l2h;
    const bool tr = t.flags & OVHIP_IF_MIP_TR;
    const uint16_t *abv = s.abv + IR_NEG, *lft = s.lft + IR_NEG;
    const int l2b = 1 << ((l2w > 2) || (l2h > 2)), nb = 1 << l2b, l2bx = l2w - l2b, l2by = l2h - l2b;
    const bool red = l2h == 2 || l2w == 2 || (l2h <= 3 && l2w <= 3);
    if (lane < 2 * nb) {
        const bool is_abv = lane < nb;
        const int j = is_abv ? lane : lane - nb, l2 = is_abv ? l2bx : l2by;
        const uint16_t *r = is_abv ? abv : lft;
        int sum = 0;
        for (int i = 0; i < (1 << l2); ++i) sum += r[1 + i + (j << l2)];
        const int v = (sum + ((1 << l2) >> 1)) >> l2;
        s.par[(is_abv != tr) ? j : nb + j] = v;                // transposed: left boundary first
    }
    __syncthreads();
    const int in_off = s.par[0];
    int bnd[8], sum = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int v = i < 2 * nb ? s.par[i] : in_off;
        if (i == 0 && red) v = 1 << (OV_BD - 1);
        bnd[i] = v - in_off; sum += i < 2 * nb ? bnd[i] : 0;
    }
    const int rnd_mip = 32 - 32 * sum;
    const int l2rw = red ? 2 : min(l2w, 3), l2rh = red ? 2 : min(l2h, 3);
    const uint8_t *mat = (l2w == 2 && l2h == 2) ? ovt_mip_4x4 + t.mode * 64 : (red ? ovt_mip_8x8 + t.mode * 128 : ovt_mip_16x16 + t.mode * 512);
    const int sx = 2 * nb, nred = 1 << (l2rw + l2rh);
    if (lane < nred) {
        int v = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) if (k < sx) v += bnd[k] * mat[lane * sx + k];
        v = ov_clip_bd(((v + rnd_mip) >> 6) + in_off);
        // reduced prediction in raster order of the (transposed back) block
        const int pos = tr ? ((lane & ((1 << l2rh) - 1)) << l2rw) + (lane >> l2rh) : lane;
        s.red[pos] = (uint16_t)v;
    }
    __syncthreads();
    const int sxs = l2w - l2rw, sys = l2h - l2rh, rw = 1 << l2rw, rh = 1 << l2rh;
    // horizontal up-sampling of the rh reduced rows (boundary = the left reference at that row), then vertical (boundary = above)
    for (int p = lane; p < rh * w; p += 64) {
        const int i = p >> l2w, x = p & (w - 1);
        int v;
        if (!sxs) v = s.red[i * rw + x];
        else {
            const int j = x >> sxs, pos = (x & ((1 << sxs) - 1)) + 1;
            const int before = j ? s.red[i * rw + j - 1] : lft[(i + 1) << sys], after = s.red[i * rw + j];
            v = (before * ((1 << sxs) - pos) + after * pos + (1 << (sxs - 1))) >> sxs;
        }
        s.hup[i * 64 + x] = (uint16_t)v;
    }
    __syncthreads();
    for (int p = lane; p < w * h; p += 64) {
        const int x = p & (w - 1), y = p >> l2w;
        int v;
        if (!sys) v = s.hup[y * 64 + x];
        else {
            const int i = y >> sys, pos = (y & ((1 << sys) - 1)) + 1;
            const int before = i ? s.hup[(i - 1) * 64 + x] : abv[1 + x], after = s.hup[i * 64 + x];
            v = (before * ((1 << sys) - pos) + after * pos + (1 << (sys - 1))) >> sys;
        }
        emit(sink, x, y, v);
    }
}

struct LmPar { int a, b, shift; };
__device__ LmPar lm_params(int min_l, int min_c, int max_c, int v, int l2rng)
{
    LmPar p;
    const int rc = max_c - min_c;
    const int l2c = rc ? ilog2(abs(rc)) + 1 : 0;
    int a = (rc * v + ((1 << l2c) >> 1)) >> l2c;
    int shift = 3 + l2rng - l2c;
    if (shift < 1) { shift = 1; a = a ? (a < 0 ? -15 : 15) : 0; }
    p.a = a; p.shift = shift; p.b = min_c - ((a * min_l) >> shift);
    return p;
}

// cross-component linear model (rcn_intra_cclm.c:56-880, the non-collocated variant)
__device__ void pred_cclm(IntraLds &s, const ovhip_pic &pic, const ovhip_itask &t, int log2_ctu, const Sink &kcb, const Sink &kcr, int lane)
{
    const int l2w = t.log2_w, w = 1 << l2w, h = 1 << t.log2_h, x0 = t.x, y0 = t.y;
    const int sl = pic.stride_y, sc = pic.stride_c;
    const uint16_t *sy = pic.y + (y0 * 2) * sl + x0 * 2, *scb = pic.cb + y0 * sc + x0, *scr = pic.cr + y0 * sc + x0;
    const int mode = t.mode;
    const bool lft_avail = t.avl_lft > 0, abv_avail = t.avl_abv > 0;
    if (lane == 0) {
        const bool first_line = !((y0 * 2) & ((1 << log2_ctu) - 1));
        int py[4], pcb[4], pcr[4], n = 0;
        int n_abv = 0, abv_step = 1, n_lft = 0, lft_step = 1;
        if (mode == 67) {
            if (abv_avail) { const int l2n = 1 + !lft_avail; abv_step = max(1, w >> l2n); n_abv = min(w, (1 + !lft_avail) << 1); }
            if (lft_avail) { const int l2n = 1 + !abv_avail; lft_step = max(1, h >> l2n); n_lft = min(h, (1 + !abv_avail) << 1); }
        } else if (mode == 69 && abv_avail) { const int len = t.avl_abv << 1; n_abv = min(len, 4); abv_step = max(1, len >> 2); }
        else if (mode == 68 && lft_avail) { const int len = t.avl_lft << 1; n_lft = min(len, 4); lft_step = max(1, len >> 2); }
        {
            const int sp = abv_step >> 1;
            const uint16_t *q = first_line ? sy - sl + (sp << 1) : sy - 2 * sl + (sp << 1);
            int pad_left = sp == 0 && !lft_avail;
            for (int i = 0; i < n_abv; ++i) {
                const int v = first_line ? (2 + q[-(!pad_left)] + 2 * q[0] + q[1]) >> 2
                                         : (4 + q[-(!pad_left)] + 2 * q[0] + q[1] + q[sl - (!pad_left)] + 2 * q[sl] + q[sl + 1]) >> 3;
                py[n] = v; pcb[n] = scb[-sc + sp + i * abv_step]; pcr[n] = scr[-sc + sp + i * abv_step]; ++n;
                q += abv_step << 1; pad_left = 0;
            }
        }
        {
            const int sp = lft_step >> 1;
            const uint16_t *q = sy - 2 + sp * 2 * sl;
            for (int i = 0; i < n_lft; ++i) {
                py[n] = (4 + 2 * q[0] + q[1] + q[-1] + 2 * q[sl] + q[sl + 1] + q[sl - 1]) >> 3;
                pcb[n] = scb[-1 + (sp + i * lft_step) * sc]; pcr[n] = scr[-1 + (sp + i * lft_step) * sc]; ++n;
                q += 2 * sl * lft_step;
            }
        }
        LmPar pb = { 0, 1 << (OV_BD - 1), 0 }, pr = { 0, 1 << (OV_BD - 1), 0 };
        if (n) {
            int min_l, max_l, min_cb, max_cb, min_cr, max_cr;
            if (n == 2) {
                const int mi = py[0] >= py[1], ma = !mi;
                min_l = py[mi]; max_l = py[ma]; min_cb = pcb[mi]; max_cb = pcb[ma]; min_cr = pcr[mi]; max_cr = pcr[ma];
            } else {
                int i0 = 0, i1 = 2, j0 = 1, j1 = 3, tt;         // (i0, i1) = minima pair, (j0, j1) = maxima pair
                if (py[i0] > py[i1]) { tt = i0; i0 = i1; i1 = tt; }
                if (py[j0] > py[j1]) { tt = j0; j0 = j1; j1 = tt; }
                if (py[i0] > py[j1]) { tt = i0; i0 = j0; j0 = tt; tt = i1; i1 = j1; j1 = tt; }
                if (py[i1] > py[j0]) { tt = i1; i1 = j0; j0 = tt; }
                min_l = (py[i0] + py[i1] + 1) >> 1; max_l = (py[j0] + py[j1] + 1) >> 1;
                min_cb = (pcb[i0] + pcb[i1] + 1) >> 1; max_cb = (pcb[j0] + pcb[j1] + 1) >> 1;
                min_cr = (pcr[i0] + pcr[i1] + 1) >> 1; max_cr = (pcr[j0] + pcr[j1] + 1) >> 1;
            }
            pb.a = 0; pb.b = min_cb; pb.shift = 0; pr.a = 0; pr.b = min_cr; pr.shift = 0;
            const int rl = max_l - min_l;
            if (rl) {
                const unsigned long long div_lut = 0x0111122334455670ull;     // {0,7,6,5,5,4,4,3,3,2,2,1,1,1,1,0}, nibble i
                int l2r = ilog2(rl);
                const int nd = ((rl << 4) >> l2r) & 15, v = (int)((div_lut >> (4 * nd)) & 15) | 8;
                l2r += nd != 0;
                pb = lm_params(min_l, min_cb, max_cb, v, l2r);
                pr = lm_params(min_l, min_cr, max_cr, v, l2r);
            }
        }
        s.par[0] = pb.a; s.par[1] = pb.b; s.par[2] = pb.shift; s.par[3] = pr.a; s.par[4] = pr.b; s.par[5] = pr.shift;
    }
    __syncthreads();
    const int a_cb = s.par[0], b_cb = s.par[1], s_cb = s.par[2], a_cr = s.par[3], b_cr = s.par[4], s_cr = s.par[5];
    for (int p = lane; p < w * h; p += 64) {
        const int i = p & (w - 1), j = p >> l2w;
        const uint16_t *q = sy + 2 * j * sl + 2 * i;
        const int pl = i == 0 && !lft_avail;
        const int v = (4 + q[1] + q[-(!pl)] + 2 * q[0] + 2 * q[sl] + q[sl + 1] + q[sl - (!pl)]) >> 3;
        emit(kcb, i, j, ov_clip_bd(((v * a_cb) >> s_cb) + b_cb));
        emit(kcr, i, j, ov_clip_bd(((v * a_cr) >> s_cr) + b_cr));
    }
}

struct LmcsWnd { uint16_t bnd[17]; int min_idx, max_idx, crs_offset; };

// rcn_lmcs_compute_chroma_scale for one region (same arithmetic as k_lmcs_scale, kernels_lmcs.hip)
__device__ void region_scale(const ovhip_pic &pic, const ovhip_lmcs_region &g, const LmcsWnd &wnd, int16_t *out, int lane)
{
    int sum = 0;
    const uint16_t *src = pic.y + (size_t)g.y * pic.stride_y + g.x;
    if (g.n_abv) sum += src[-pic.stride_y + min(lane, 4 * g.n_abv - 1)];
    if (g.n_lft) sum += src[(size_t)min(lane, 4 * g.n_lft - 1) * pic.stride_y - 1];
#pragma unroll
    for (int m = 32; m; m >>= 1) sum += __shfl_xor(sum, m);
    if (lane == 0) {
        const int nb_units = (g.n_abv ? 16 : 0) + (g.n_lft ? 16 : 0);
        int log2_nb = 0;
        for (int v = nb_units; v; v >>= 1) ++log2_nb;
        const int avg = log2_nb ? (sum + (1 << log2_nb)) >> (log2_nb + 1) : 512;
        int idx = wnd.min_idx;
        for (; idx < wnd.max_idx; ++idx) if (avg < wnd.bnd[idx + 1]) break;
        idx = min(idx, 15);
        const int wnd_sz = (int)wnd.bnd[idx + 1] - (int)wnd.bnd[idx];
        *out = (int16_t)(wnd_sz == 0 ? 1 << 11 : (1 << (OV_BD - 4 + 11)) / (wnd_sz + wnd.crs_offset));
    }
}

__global__ __launch_bounds__(64) void k_intra_level(ovhip_pic pic, ovhip_pic res, const ovhip_itask *__restrict__ tasks, uint32_t n,
                                                    const ovhip_lmcs_region *__restrict__ regs, LmcsWnd wnd, int16_t *__restrict__ scales,
                                                    int log2_ctu)
{
    __shared__ IntraLds s;
    const uint32_t bid = blockIdx.x;
    if (bid >= n) return;
    const ovhip_itask t = tasks[bid];
    const int lane = threadIdx.x;
    const int w = 1 << t.log2_w, h = 1 << t.log2_h;
    const bool scaled = t.flags & OVHIP_IF_RES_SCALE;
    const int scale = scaled ? ((t.flags & OVHIP_IF_SCALE_IDX) ? scales[t.c_scale] : t.c_scale) : 0;
    if (t.kind == OVHIP_IT_REGION) { region_scale(pic, regs[t.c_scale], wnd, scales + t.c_scale, lane); return; }
    if (t.kind == OVHIP_IT_LUMA) {
        Sink k;
        k.dst = pic.y + t.y * pic.stride_y + t.x; k.dstride = pic.stride_y;
        k.res = (t.flags & OVHIP_IF_RES_Y) ? reinterpret_cast<const int16_t *>(res.y) + t.y * res.stride_y + t.x : nullptr; k.rstride = res.stride_y;
        k.scaled = 0; k.scale = 0; k.ciip_wt = t.ciip_wt;
        fetch_refs(s, pic.y, pic.stride_y, t.x, t.y, w, h, 4, t.flags & OVHIP_IF_CORNER, t.avl_abv, t.avl_lft, (t.flags & OVHIP_IF_MIP) ? 0 : t.mrl_idx, lane);
        __syncthreads();
        if (t.flags & OVHIP_IF_MIP) pred_mip(s, t, k, lane);
        else pred_regular(s, t, true, k, lane);
        return;
    }
    // chroma: Cb then Cr
    Sink kcb, kcr;
    kcb.dst = pic.cb + t.y * pic.stride_c + t.x; kcr.dst = pic.cr + t.y * pic.stride_c + t.x; kcb.dstride = kcr.dstride = pic.stride_c;
    kcb.res = (t.flags & OVHIP_IF_RES_CB) ? reinterpret_cast<const int16_t *>(res.cb) + t.y * res.stride_c + t.x : nullptr;
    kcr.res = (t.flags & OVHIP_IF_RES_CR) ? reinterpret_cast<const int16_t *>(res.cr) + t.y * res.stride_c + t.x : nullptr;
    kcb.rstride = kcr.rstride = res.stride_c;
    kcb.scaled = kcr.scaled = scaled; kcb.scale = kcr.scale = scale; kcb.ciip_wt = kcr.ciip_wt = t.ciip_wt;
    if (t.kind == OVHIP_IT_RES_C) {
        // residual of an already predicted block: prediction = what is there
        for (int p = lane; p < w * h; p += 64) {
            const int x = p & (w - 1), y = p >> t.log2_w;
            if (kcb.res) { Sink q = kcb; q.ciip_wt = 0; emit(q, x, y, kcb.dst[y * kcb.dstride + x]); }
            if (kcr.res) { Sink q = kcr; q.ciip_wt = 0; emit(q, x, y, kcr.dst[y * kcr.dstride + x]); }
        }
        return;
    }
    if (t.mode >= 67) { pred_cclm(s, pic, t, log2_ctu, kcb, kcr, lane); return; }
    fetch_refs(s, pic.cb, pic.stride_c, t.x, t.y, w, h, 2, t.flags & OVHIP_IF_CORNER, t.avl_abv, t.avl_lft, 0, lane);
    __syncthreads();
    pred_regular(s, t, false, kcb, lane);
    __syncthreads();
    fetch_refs(s, pic.cr, pic.stride_c, t.x, t.y, w, h, 2, t.flags & OVHIP_IF_CORNER, t.avl_abv, t.avl_lft, 0, lane);
    __syncthreads();
    pred_regular(s, t, false, kcr, lane);
}

} // namespace

// One level of the ordered pass.  d_tasks: DEVICE, the n tasks of this level.  res: residual picture written by
// ovhip_itx_launch_classes_res.  d_regions / luts / d_scales: as ovhip_lmcs_scale_launch (may be NULL without LMCS chroma scaling).
extern "C" int ovhip_intra_level_launch(ovhip_ctx *ctx, const ovhip_pic *pic, const ovhip_pic *res, const ovhip_itask *d_tasks, uint32_t n,
                                        const ovhip_lmcs_region *d_regions, const ovhip_lmcs_luts *luts, int16_t *d_scales, int32_t log2_ctu_s)
{
    if (!ctx || !pic || !res) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    if (!n) return OVHIP_OK;
    if (!d_tasks || log2_ctu_s < 5 || log2_ctu_s > 7) return ov_fail(ctx, OVHIP_EINVAL, "ovhip_intra_level_launch: bad arguments", hipSuccess);
    LmcsWnd wnd;
    memset(&wnd, 0, sizeof(wnd));
    if (luts) { memcpy(wnd.bnd, luts->wnd_bnd, sizeof(wnd.bnd)); wnd.min_idx = luts->min_idx; wnd.max_idx = luts->max_idx; wnd.crs_offset = luts->crs_offset; }
    hipLaunchKernelGGL(k_intra_level, dim3(n), dim3(64), 0, ctx->stream, *pic, *res, d_tasks, n, d_regions, wnd, d_scales, log2_ctu_s);
    OV_LAUNCH_CHECK(ctx, "k_intra_level");
    return OVHIP_OK;
}
