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
// Mapping: one 64-lane workgroup per (task, strip of 1024 samples, colour plane); lane = sample (<= 16 samples per lane).
// A level is LATENCY bound (few, small, mutually independent tasks; the next level waits for all of them), so the kernel is
// arranged around the number of dependent memory round trips: task -> {reference samples, residual, CIIP's inter samples:
// all issued together} -> prediction out of LDS into an LDS tile -> epilogue (blend, residual, clip, store).
// Reference samples of the block are fetched once into LDS with the substitution rules of 8.4.5.2.8 (availability = unit
// counts from the recorder), smoothed copies next to them when the mode asks for them; every prediction mode then reads
// LDS only.  int16 / int32 arithmetic, no MFMA: per-sample stencils.
#include <stdlib.h>
#include "ovvc_common.hip.h"
#include "flow_state.hip.h"
#define OVT_ATTR __device__
#include "vvc_mip_tables.h"

namespace {

__device__ const short g_ang[32] = { 0, 1, 2, 3, 4, 6, 8, 10, 12, 14, 16, 18, 20, 23, 26, 29, 32, 35, 39, 45, 51, 57, 64, 73, 86, 102, 128, 171, 256, 341, 512, 1024 };
__device__ const short g_inv_ang[32] = { 0, 16384, 8192, 5461, 4096, 2731, 2048, 1638, 1365, 1170, 1024, 910, 819, 712, 630, 565, 512, 468, 420, 364,
                                         321, 287, 256, 224, 191, 161, 128, 96, 64, 48, 32, 16 };
__device__ __attribute__((aligned(4))) const signed char g_fc[32][4] = {
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
    int      par[16];                    // CCLM neighbour samples / MIP boundary
    short    ang[32], inv_ang[32];       // the angle tables, staged by load_tables() next to the task load (not behind it)
    signed char fc[32][4];
    __attribute__((aligned(16))) uint16_t pred[1024];               // the strip's predicted samples
};
#define STRIP 1024
#define NPL   (STRIP / 64)

struct Strip { int p0, p1; };            // raster sample range [p0, p1) of the block this workgroup predicts

// One memory round trip less on the critical path of a task: the tables do not depend on the task, so they are requested
// before it is known which entries will be needed.
__device__ __forceinline__ void load_tables(IntraLds &s, int lane)
{
    if (lane < 32) {
        s.ang[lane] = g_ang[lane]; s.inv_ang[lane] = g_inv_ang[lane];
        *reinterpret_cast<int *>(s.fc[lane]) = *reinterpret_cast<const int *>(g_fc[lane]);
    }
}

__device__ __forceinline__ int pdpc_wgt(int i, int scale) { const int sh = (i << 1) >> scale; return sh > 5 ? 0 : 32 >> sh; }
__device__ __forceinline__ int ilog2(int v) { return 31 - __clz(v); }

// Where the reconstructed samples of a plane are read from, in the plane's picture coordinates: the picture itself (level
// kernel) or the CTU tile in LDS with its borders (CTU kernel; row -1 of the tile lives in `top`, which is longer: above-right).
struct PlaneAcc {
    const uint16_t *p; int stride;
    __device__ __forceinline__ int ld(int x, int y) const { return p[y * stride + x]; }
};
struct TileAcc {
    const uint16_t *tile, *top; int stride, ox, oy;          // tile[0] = sample (ox - 4, oy); top[0] = sample (ox - 4, oy - 1)
    __device__ __forceinline__ int ld(int x, int y) const { x -= ox - 4; y -= oy; return y < 0 ? top[x] : tile[y * stride + x]; }
};

// all cross-lane traffic of a task goes through the LDS of ONE wave: program order of the wave's DS instructions is enough
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// reference samples with substitution: see fetch_refs in oracle/ovvc_oracle_intra.c for the sequential form; with the availability
// the recorder hands over (corner flag + unit counts from the corner outwards) every sample's source is known in closed form
template <class Acc>
__device__ __forceinline__ void fetch_refs(IntraLds &s, const Acc acc, int x0, int y0, int w, int h, int unit, bool corner,
                                           int avl_abv, int avl_lft, int mrl, int lane)
{
    const int na = 2 * w + mrl + 1, nl = 2 * h + mrl + 1;
    const int l2u = unit == 4 ? 2 : 1;                                           // units are 4 luma / 2 chroma samples
    const int cx = x0 - 1 - mrl, cy = y0 - 1 - mrl;                               // sample (k = 0) of both arms
    // Every reference sample, substituted or not, is a copy of ONE picture sample: pick its coordinates first, then load -- all
    // loads of both arms are independent (one memory round trip on the critical path of an ordered task instead of two).
    const bool none = !corner && !avl_abv && !avl_lft;
    const int la = min(mrl + avl_abv * unit, na - 1), ll = min(mrl + avl_lft * unit, nl - 1);   // last available sample per arm
    const int ax1 = cx + mrl + 1, ay1 = cy, lx1 = cx, ly1 = cy + mrl + 1;   // first sample of each arm's block part
    // (na + 24 and nl + 24 are at most 2 * 64 + 4 + 24 = 156: three samples per lane and arm.)  ALL loads of both arms go out before the
    // first one is waited for -- a loop that loads, waits and stores per iteration costs one memory round trip per iteration and arm
    // Lanes past the end of an arm load its last sample again: every load is unconditional (a load inside a divergent branch is
    // waited for at the branch's end, which serialises the round trips), the coordinates are selected without branches.
    int va[3], vl[3];
    if (none) {
#pragma unroll
        for (int i = 0; i < 3; ++i) va[i] = vl[i] = 1 << (OV_BD - 1);
    } else {
        const bool fa0 = (mrl == 0 && avl_abv && avl_lft) || !avl_lft;                 // corner part of the above arm without a corner: whose first sample
        const int cax = corner ? 0 : (fa0 ? ax1 : lx1), cay = fa0 ? ay1 : ly1;           // (x used only when !corner)
        const int eax = avl_abv ? cx + la : (corner ? cx + mrl : lx1), eay = (avl_abv || corner) ? cy : ly1;     // past the available part
        const int clx = avl_lft ? lx1 : ax1, cly = avl_lft ? ly1 : ay1;
        const int elx = (avl_lft || corner) ? cx : ax1, ely = avl_lft ? cy + ll : (corner ? cy + mrl : ay1);
        int ax_[3], ay_[3], lx_[3], ly_[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int k = lane + 64 * i;
            const int ka = min(k, na - 1), kl = min(k, nl - 1);
            const bool a_corner = ka <= mrl, a_avail = ((ka - mrl - 1) >> l2u) < avl_abv;
            ax_[i] = a_corner ? (corner ? cx + ka : cax) : (a_avail ? cx + ka : eax);
            ay_[i] = a_corner ? (corner ? cy : cay) : (a_avail ? cy : eay);
            const bool l_corner = kl <= mrl, l_avail = ((kl - mrl - 1) >> l2u) < avl_lft;
            lx_[i] = l_corner ? (corner ? cx : clx) : (l_avail ? cx : elx);
            ly_[i] = l_corner ? (corner ? cy + kl : cly) : (l_avail ? cy + kl : ely);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) { va[i] = acc.ld(ax_[i], ay_[i]); vl[i] = acc.ld(lx_[i], ly_[i]); }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int k = lane + 64 * i;
        if (k < na + 24) s.abv[IR_NEG + k] = (uint16_t)va[i];
        if (k < nl + 24) s.lft[IR_NEG + k] = (uint16_t)vl[i];
    }
}

// Reference arms of an ISP prediction call (OVHIP_IF_ISP): the CODING UNIT's arms as fill_ref_above_0 / fill_ref_left_0 build
// them (rcn_fill_ref.c:71-150, :320-390), shifted by the block's offset and cut at cb + pb samples (intra_pred_isp,
// rcn_intra.c:584-603).  Per arm and in closed form: S(k) = sample k steps from the CU arm's corner, c = corner unit available in
// this arm's map, a = units available behind it, other = the other arm has anything, F = the other arm's first sample here.
template <class Acc>
__device__ __forceinline__ void fetch_refs_isp(IntraLds &s, const Acc acc, const ovhip_itask &t, int lane)
{
    const int w = 1 << t.log2_w, h = 1 << t.log2_h, cbw = 1 << t.isp_log2_cb_w, cbh = 1 << t.isp_log2_cb_h;
    const int ox = t.isp_off_x, oy = t.isp_off_y, x0 = t.x, y0 = t.y;
    const bool ca = t.flags & OVHIP_IF_CORNER, cl = t.flags & OVHIP_IF_CORNER_L;
    const int aa = t.avl_abv, al = t.avl_lft;
    for (int arm = 0; arm < 2; ++arm) {
        const int cb = arm ? cbh : cbw, pb = arm ? h : w, off = arm ? oy : ox, a = arm ? al : aa;
        const bool c = arm ? cl : ca, other = arm ? (ca || aa) : (cl || al);
        const int n_cu = 2 * cb + 1, nb_ref = (2 * cb) / 4 + 1, len = cb + pb, last = min(4 * a, n_cu + 3);
        uint16_t *out = (arm ? s.lft : s.abv) + IR_NEG;
        for (int k = lane; k < len + 1 + 24; k += 64) {
            const int kk = min(k, len) + off;
            int v;
            if (c && a >= nb_ref) {
                const int q = min(kk, n_cu - 1);
                v = arm ? acc.ld(x0 - 1, y0 - oy - 1 + q) : acc.ld(x0 - ox - 1 + q, y0 - 1);
            } else if (c || a) {
                const int q = kk == 0 ? (c ? 0 : 1) : (a ? min(kk, last) : 0);
                v = arm ? acc.ld(x0 - 1, y0 - oy - 1 + q) : acc.ld(x0 - ox - 1 + q, y0 - 1);
            } else {
                v = other ? (arm ? acc.ld(x0, y0 - 1) : acc.ld(x0 - 1, y0)) : 1 << (OV_BD - 1);
            }
            out[k] = (uint16_t)v;
        }
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

__device__ __forceinline__ int res_scale(int v, int scale)
{
    const int sign = v & (1 << 15);
    int a = (ov_clip_bd(abs(v)) * scale + (1 << 10)) >> 11;
    return ov_clip3(sign ? -a : a, -(1 << 15), 1 << 15);
}

// planar / DC / angular / BDPCM prediction of one plane's block out of the LDS references
__device__ void pred_regular(IntraLds &s, const ovhip_itask &t, bool is_luma, const Strip st, int lane)
{
    const int l2w = t.log2_w, l2h = t.log2_h, w = 1 << l2w, h = 1 << l2h, n = w * h;
    const bool isp = t.flags & OVHIP_IF_ISP;          // always cubic, never smoothed, PDPC from 4 rows on, wide angles by the CU
    const int mrl = (is_luma && !isp) ? t.mrl_idx : 0;
    const bool bdpcm = t.flags & OVHIP_IF_BDPCM;
    const bool pdpc_ok = isp ? l2h > 1 : (!mrl && !bdpcm && (is_luma || (l2w > 1 && l2h > 1)));
    const uint16_t *abv = s.abv + IR_NEG, *lft = s.lft + IR_NEG;
    if (bdpcm) {
        const bool ver = t.flags & OVHIP_IF_BDPCM_VER;
        for (int p = st.p0 + lane; p < st.p1; p += 64) { const int x = p & (w - 1), y = p >> l2w; s.pred[p - st.p0] = ver ? abv[1 + x] : lft[1 + y]; }
        return;
    }
    int mode = t.mode;
    if (mode == 0) {
        const uint16_t *a = abv + mrl, *l = lft + mrl;
        if (is_luma && !isp && !mrl && l2w + l2h > 5) { smooth_refs(s, w + 4, h + 4, lane); wave_sync(); a = s.fabv + IR_NEG; l = s.flft + IR_NEG; }
        const int scale = (l2w + l2h - 2) >> 2;
        for (int p = st.p0 + lane; p < st.p1; p += 64) {
            const int x = p & (w - 1), y = p >> l2w;
            const int pv = ((h - 1 - y) * a[1 + x] + (y + 1) * l[1 + h]) << l2w;
            const int ph = ((w - 1 - x) * l[1 + y] + (x + 1) * a[1 + w]) << l2h;
            int v = (pv + ph + n) >> (l2w + l2h + 1);
            if (pdpc_ok) { const int wt = pdpc_wgt(y, scale), wl = pdpc_wgt(x, scale); v = ov_clip_bd((l[1 + y] * wl + a[1 + x] * wt + (64 - wl - wt) * v + 32) >> 6); }
            s.pred[p - st.p0] = (uint16_t)v;
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
        for (int p = st.p0 + lane; p < st.p1; p += 64) {
            const int x = p & (w - 1), y = p >> l2w;
            int v = dc;
            if (pdpc_ok) { const int wt = pdpc_wgt(y, scale), wl = pdpc_wgt(x, scale); v = ov_clip_bd((l[1 + y] * wl + a[1 + x] * wt + (64 - wl - wt) * v + 32) >> 6); }
            s.pred[p - st.p0] = (uint16_t)v;
        }
        return;
    }
    // wide-angle remap (derive_wide_angular_mode, rcn_intra.c:54-66; the reference's numbering below mode 2)
    {
        const int sw = isp ? t.isp_log2_cb_w : l2w, sh = isp ? t.isp_log2_cb_h : l2h;
        if (sw != sh) {
            const int r = abs(sw - sh);
            if (sw > sh && mode < (r > 1 ? 8 + 2 * r : 8)) mode += 65;
            else if (sh > sw && mode > (r > 1 ? 60 - 2 * r : 60)) mode -= 65;
        }
    }
    const bool vertical = mode >= 34;
    const int midx = vertical ? mode - 50 : 18 - mode, am = abs(midx);
    const int angle_abs = s.ang[am], inv = s.inv_ang[am];
    const int angle = midx < 0 ? -angle_abs : angle_abs;
    bool use_fg = false, smoothed = false;
    if (is_luma && !isp && !mrl && l2w + l2h > 5 && am > (int)((0x000000020E181818ull >> (8 * ((l2w + l2h) >> 1))) & 0xff)) {
        if (!(angle_abs & 31)) { smooth_refs(s, 2 * w, 2 * h, lane); wave_sync(); smoothed = true; }
        else use_fg = true;
    }
    uint16_t *mainr = (vertical ? (smoothed ? s.fabv : s.abv) : (smoothed ? s.flft : s.lft)) + IR_NEG;
    const uint16_t *side = (vertical ? (smoothed ? s.flft : s.lft) : (smoothed ? s.fabv : s.abv)) + IR_NEG;
    const int mw = vertical ? w : h, mh = vertical ? h : w, l2mh = vertical ? l2h : l2w;     // block in the mode's own orientation
    if (midx < 0) {
        for (int k = 1 + lane; k <= mh; k += 64) { int si = (256 + k * inv) >> 9; si = min(si, mh); mainr[-k] = side[si]; }
        wave_sync();
    }
    int nscale = -1;
    if (pdpc_ok) {
        if (am == 0) nscale = (l2w + l2h - 2) >> 2;
        else if (midx > 0) nscale = min(2, l2mh - (ilog2(3 * inv - 2) - 8));
    }
    const int tl = s.abv[IR_NEG];                         // pure-direction PDPC subtracts the ABOVE array's corner (rcn_intra_angular.c:308, :328)
    for (int p = st.p0 + lane; p < st.p1; p += 64) {
        const int x = p & (w - 1), y = p >> l2w;
        const int mx = vertical ? x : y, my = vertical ? y : x;     // position in the mode's orientation
        const int pos = (my + 1 + mrl) * angle;
        const int iidx = (pos >> 5) + mrl, ifact = pos & 31;
        const uint16_t *r = mainr + mx + iidx;
        int v;
        if (is_luma) {
            if (!(angle_abs & 31)) v = r[1];
            else {
                int f0, f1, f2, f3;
                if (use_fg) { f0 = 16 - (ifact >> 1); f1 = 32 - (ifact >> 1); f2 = 16 + (ifact >> 1); f3 = ifact >> 1; }
                else { f0 = s.fc[ifact][0]; f1 = s.fc[ifact][1]; f2 = s.fc[ifact][2]; f3 = s.fc[ifact][3]; }
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
        s.pred[p - st.p0] = (uint16_t)v;
    }
}

// matrix-based intra prediction (rcn_intra_mip.c:44-400)
__device__ void pred_mip(IntraLds &s, const ovhip_itask &t, const Strip st, int lane)
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
    wave_sync();
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
    // the lane's matrix row as two dwords, unconditionally (lanes past the reduced block read row 0; a 4-column row reads its dword
    // twice): byte loads inside `if (k < sx)` were four round trips one after the other
    const uint8_t *mrow = mat + (lane < nred ? lane : 0) * sx;
    const uint32_t m0 = *reinterpret_cast<const uint32_t *>(mrow), m1 = *reinterpret_cast<const uint32_t *>(mrow + (sx == 8 ? 4 : 0));
    if (lane < nred) {
        int v = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) if (k < sx) v += bnd[k] * (int)(((k < 4 ? m0 : m1) >> (8 * (k & 3))) & 0xff);
        v = ov_clip_bd(((v + rnd_mip) >> 6) + in_off);
        // reduced prediction in raster order of the (transposed back) block
        const int pos = tr ? ((lane & ((1 << l2rh) - 1)) << l2rw) + (lane >> l2rh) : lane;
        s.red[pos] = (uint16_t)v;
    }
    wave_sync();
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
    wave_sync();
    for (int p = st.p0 + lane; p < st.p1; p += 64) {
        const int x = p & (w - 1), y = p >> l2w;
        int v;
        if (!sys) v = s.hup[y * 64 + x];
        else {
            const int i = y >> sys, pos = (y & ((1 << sys) - 1)) + 1;
            const int before = i ? s.hup[(i - 1) * 64 + x] : abv[1 + x], after = s.hup[i * 64 + x];
            v = (before * ((1 << sys) - pos) + after * pos + (1 << (sys - 1))) >> sys;
        }
        s.pred[p - st.p0] = (uint16_t)v;
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

// cross-component linear model (rcn_intra_cclm.c:56-880, the non-collocated variant), one chroma plane.  The up to four
// neighbour positions are sampled by lanes 0..3 in parallel (above positions first, as the reference orders them).
// NP: samples per lane of a strip (the caller's strip size / 64).  All loads -- the block's down-sampled luma and the
// neighbour taps the parameters come from -- are issued before the first one is waited for, none of them inside a divergent branch.
template <int NP, class Acc>
__device__ __forceinline__ void pred_cclm(IntraLds &s, const Acc ya, const Acc ca, const ovhip_itask &t, int log2_ctu, const Strip st, int lane)
{
    const int l2w = t.log2_w, w = 1 << l2w, h = 1 << t.log2_h, x0 = t.x, y0 = t.y;
#define Y(dx, dy) ya.ld(2 * x0 + (dx), 2 * y0 + (dy))
    const int mode = t.mode;
    const bool lft_avail = t.avl_lft > 0, abv_avail = t.avl_abv > 0;
    const bool first_line = !((y0 * 2) & ((1 << log2_ctu) - 1));
    int n_abv = 0, abv_step = 1, n_lft = 0, lft_step = 1;
    if (mode == 67) {
        if (abv_avail) { const int l2n = 1 + !lft_avail; abv_step = max(1, w >> l2n); n_abv = min(w, (1 + !lft_avail) << 1); }
        if (lft_avail) { const int l2n = 1 + !abv_avail; lft_step = max(1, h >> l2n); n_lft = min(h, (1 + !abv_avail) << 1); }
    } else if (mode == 69 && abv_avail) { const int len = t.avl_abv << 1; n_abv = min(len, 4); abv_step = max(1, len >> 2); }
    else if (mode == 68 && lft_avail) { const int len = t.avl_lft << 1; n_lft = min(len, 4); lft_step = max(1, len >> 2); }
    const int n = n_abv + n_lft;
    // the strip's down-sampled luma (lanes past the end of the strip repeat its last sample)
    int dsv[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int p = min(st.p0 + lane + 64 * q, st.p1 - 1);
        const int i = p & (w - 1), j = p >> l2w;
        const int pl = i == 0 && !lft_avail;
        dsv[q] = (4 + Y(2 * i + 1, 2 * j) + Y(2 * i - !pl, 2 * j) + 2 * Y(2 * i, 2 * j) + 2 * Y(2 * i, 2 * j + 1) + Y(2 * i + 1, 2 * j + 1)
                  + Y(2 * i - !pl, 2 * j + 1)) >> 3;
    }
    if (n) {
        // neighbour `ln` (lanes >= n repeat the last one): six luma taps with weights 1 2 1 / 1 2 1 -- above: columns qx - 1 .. qx + 1
        // of rows -2 and -1 (the CTU's first line has row -1 only: taken twice, the same value); left: columns -3 .. -1 of rows qy,
        // qy + 1 -- and the chroma sample beside the block
        const int ln = min(lane, n - 1);
        const bool is_abv = ln < n_abv;
        const int pos = is_abv ? (abv_step >> 1) + ln * abv_step : (lft_step >> 1) + (ln - n_abv) * lft_step;
        const int q2 = pos << 1, pl = pos == 0 && !lft_avail;
        const int r2 = first_line ? -1 : -2;
        const int x0_ = is_abv ? q2 - !pl : -3, x1_ = is_abv ? q2 : -2, x2_ = is_abv ? q2 + 1 : -1;
        const int ya_ = is_abv ? r2 : q2, yb_ = is_abv ? -1 : q2 + 1;
        const int v = (4 + Y(x0_, ya_) + 2 * Y(x1_, ya_) + Y(x2_, ya_) + Y(x0_, yb_) + 2 * Y(x1_, yb_) + Y(x2_, yb_)) >> 3;
        const int c = ca.ld(is_abv ? x0 + pos : x0 - 1, is_abv ? y0 - 1 : y0 + pos);
        if (lane < n) { s.par[lane] = v; s.par[4 + lane] = c; }
    }
    wave_sync();
    LmPar pp = { 0, 1 << (OV_BD - 1), 0 };
    if (n) {
        int py[4], pc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { py[k] = s.par[k]; pc[k] = s.par[4 + k]; }
        int min_l, max_l, min_c, max_c;
        if (n == 2) {
            const bool sw = py[0] >= py[1];
            min_l = sw ? py[1] : py[0]; max_l = sw ? py[0] : py[1]; min_c = sw ? pc[1] : pc[0]; max_c = sw ? pc[0] : pc[1];
        } else {
            // the reference's four-sample network: (i0, i1) = minima pair, (j0, j1) = maxima pair, carried as values
            int l0 = py[0], l1 = py[2], m0 = py[1], m1 = py[3], c0 = pc[0], c1 = pc[2], d0 = pc[1], d1 = pc[3], tt;
            if (l0 > l1) { tt = l0; l0 = l1; l1 = tt; tt = c0; c0 = c1; c1 = tt; }
            if (m0 > m1) { tt = m0; m0 = m1; m1 = tt; tt = d0; d0 = d1; d1 = tt; }
            if (l0 > m1) { tt = l0; l0 = m0; m0 = tt; tt = c0; c0 = d0; d0 = tt; tt = l1; l1 = m1; m1 = tt; tt = c1; c1 = d1; d1 = tt; }
            if (l1 > m0) { tt = l1; l1 = m0; m0 = tt; tt = c1; c1 = d0; d0 = tt; }
            min_l = (l0 + l1 + 1) >> 1; max_l = (m0 + m1 + 1) >> 1; min_c = (c0 + c1 + 1) >> 1; max_c = (d0 + d1 + 1) >> 1;
        }
        pp.a = 0; pp.b = min_c; pp.shift = 0;
        const int rl = max_l - min_l;
        if (rl) {
            const unsigned long long div_lut = 0x0111122334455670ull;     // {0,7,6,5,5,4,4,3,3,2,2,1,1,1,1,0}, nibble i
            int l2r = ilog2(rl);
            const int nd = ((rl << 4) >> l2r) & 15, v = (int)((div_lut >> (4 * nd)) & 15) | 8;
            l2r += nd != 0;
            pp = lm_params(min_l, min_c, max_c, v, l2r);
        }
    }
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int p = st.p0 + lane + 64 * q;
        if (p < st.p1) s.pred[p - st.p0] = (uint16_t)ov_clip_bd(((dsv[q] * pp.a) >> pp.shift) + pp.b);
    }
#undef Y
}

struct LmcsWnd { uint16_t bnd[17]; int min_idx, max_idx, crs_offset; };

// rcn_lmcs_compute_chroma_scale for one region (same arithmetic as k_lmcs_scale, kernels_lmcs.hip)
template <class Acc>
__device__ __forceinline__ int region_scale(const Acc ya, const ovhip_lmcs_region &g, const LmcsWnd &wnd, int lane)
{
    // both loads unconditional (see k_lmcs_scale): a side that does not exist reads the region's own first sample and is dropped
    const int va = ya.ld(g.n_abv ? g.x + min(lane, 4 * g.n_abv - 1) : g.x, g.n_abv ? g.y - 1 : g.y);
    const int vl = ya.ld(g.n_lft ? g.x - 1 : g.x, g.n_lft ? g.y + min(lane, 4 * g.n_lft - 1) : g.y);
    int sum = (g.n_abv ? va : 0) + (g.n_lft ? vl : 0);
#pragma unroll
    for (int m = 32; m; m >>= 1) sum += __shfl_xor(sum, m);
    const int nb_units = (g.n_abv ? 16 : 0) + (g.n_lft ? 16 : 0);
    int log2_nb = 0;
    for (int v = nb_units; v; v >>= 1) ++log2_nb;
    const int avg = log2_nb ? (sum + (1 << log2_nb)) >> (log2_nb + 1) : 512;
    int idx = wnd.min_idx;
    for (; idx < wnd.max_idx; ++idx) if (avg < wnd.bnd[idx + 1]) break;
    idx = min(idx, 15);
    const int wnd_sz = (int)wnd.bnd[idx + 1] - (int)wnd.bnd[idx];
    return wnd_sz == 0 ? 1 << 11 : (1 << (OV_BD - 4 + 11)) / (wnd_sz + wnd.crs_offset);
}

__global__ __launch_bounds__(64) void k_intra_level(ovhip_pic pic, ovhip_pic res, const ovhip_itask *__restrict__ tasks, uint32_t n,
                                                    const ovhip_lmcs_region *__restrict__ regs, LmcsWnd wnd, int16_t *__restrict__ scales,
                                                    int log2_ctu)
{
    __shared__ IntraLds s;
    const uint32_t bid = blockIdx.x;
    const int strip = blockIdx.y, comp = blockIdx.z;
    if (bid >= n) return;
    const int lane = threadIdx.x;
    load_tables(s, lane);
    const ovhip_itask t = tasks[bid];
    const int l2w = t.log2_w, w = 1 << l2w, h = 1 << t.log2_h, npx = w * h;
    const PlaneAcc ya = { pic.y, pic.stride_y };
    if (t.kind == OVHIP_IT_REGION) {
        if (!strip && !comp) { const int v = region_scale(ya, regs[t.c_scale], wnd, lane); if (lane == 0) scales[t.c_scale] = (int16_t)v; }
        return;
    }
    const bool luma = t.kind == OVHIP_IT_LUMA;
    Strip st; st.p0 = strip * STRIP; st.p1 = min(npx, st.p0 + STRIP);
    if (st.p0 >= npx || (luma && comp)) return;
    const bool has_res = t.flags & (luma ? OVHIP_IF_RES_Y : (comp ? OVHIP_IF_RES_CR : OVHIP_IF_RES_CB));
    const bool res_only = t.kind == OVHIP_IT_RES_C;
    if (res_only && !has_res) return;
    uint16_t *pl = luma ? pic.y : (comp ? pic.cr : pic.cb);
    const int dstride = luma ? pic.stride_y : pic.stride_c, rstride = luma ? res.stride_y : res.stride_c;
    uint16_t *dst = pl + t.y * dstride + t.x;
    const int16_t *rp = reinterpret_cast<const int16_t *>(luma ? res.y : (comp ? res.cr : res.cb)) + t.y * rstride + t.x;
    const int ciip_wt = res_only ? 0 : t.ciip_wt;
    const bool need_d = ciip_wt || res_only;

    // ISP blocks of several thin partitions: only those that carry a residual add one
    const int res_mask = (t.flags & OVHIP_IF_ISP) ? t.isp_res_mask : 0xff, res_l2pb = (t.flags & OVHIP_IF_ISP) ? t.isp_log2_pb : 6;
    // everything the epilogue needs from memory is requested before the prediction starts
    int rv[NPL], dv[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int p = st.p0 + lane + 64 * i, x = p & (w - 1), y = p >> l2w;
        rv[i] = (has_res && p < st.p1 && ((res_mask >> (x >> res_l2pb)) & 1)) ? rp[y * rstride + x] : 0;
        dv[i] = (need_d && p < st.p1) ? dst[y * dstride + x] : 0;
    }
    const bool scaled = !luma && (t.flags & OVHIP_IF_RES_SCALE);
    const int scale = scaled ? ((t.flags & OVHIP_IF_SCALE_IDX) ? scales[t.c_scale] : t.c_scale) : 0;

    if (!res_only) {
        if (luma) {
            if (t.flags & OVHIP_IF_ISP) fetch_refs_isp(s, ya, t, lane);
            else fetch_refs(s, ya, t.x, t.y, w, h, 4, t.flags & OVHIP_IF_CORNER, t.avl_abv, t.avl_lft, (t.flags & OVHIP_IF_MIP) ? 0 : t.mrl_idx, lane);
            wave_sync();
            if (t.flags & OVHIP_IF_MIP) pred_mip(s, t, st, lane);
            else pred_regular(s, t, true, st, lane);
        } else {
            const PlaneAcc ca = { pl, pic.stride_c };
            if (t.mode >= 67) pred_cclm<NPL>(s, ya, ca, t, log2_ctu, st, lane);
            else {
                fetch_refs(s, ca, t.x, t.y, w, h, 2, t.flags & OVHIP_IF_CORNER, t.avl_abv, t.avl_lft, 0, lane);
                wave_sync();
                pred_regular(s, t, false, st, lane);
            }
        }
        wave_sync();
    }
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int p = st.p0 + lane + 64 * i, x = p & (w - 1), y = p >> l2w;
        if (p >= st.p1) break;
        int v = res_only ? dv[i] : s.pred[p - st.p0];
        if (ciip_wt) v = (v * ciip_wt + dv[i] * (4 - ciip_wt) + 2) >> 2;
        if (has_res) v = ov_clip_bd(v + (scaled ? res_scale(rv[i], scale) : rv[i]));
        dst[y * dstride + x] = (uint16_t)v;
    }
}


// ------------------------------------------------------------------------------------------------------------------------
// The ordered pass as ONE launch: a workgroup per CTU that holds ordered tasks, the CTU's samples in LDS.
//
// Inside a CTU the dependent hops (1809 levels across a 4K I picture, 28 per CTU) cost an LDS round trip instead of a launch
// boundary; between CTUs the reference's own wavefront rule applies (a CTU needs its left, above-left, above and above-right
// neighbours: ctudec's WPP, ovthreads / slicedec), carried by one flag word per CTU:
//   producer: tile stored write-through (agent-scope 8-byte stores), every wave drains, barrier, ONE relaxed agent flag store;
//   consumer: lanes 0..3 of wave 0 poll their neighbour's flag (relaxed, agent, s_sleep, BOUNDED), one agent acquire, barrier,
//             plain loads.
// A workgroup waits only for workgroups with a lower index (the list is in raster order); the hardware starts workgroups in
// index order, but nothing promises it, so the wait is bounded: on expiry the picture's abort word is set, every other
// waiter leaves too, and the host reports OVHIP_EHIP -- never a hang.
#define CT_S     128                        // largest CTU
#define CT_YS    (4 + CT_S)                 // luma tile row: 4 border samples + the CTU
#define CT_CS    (4 + CT_S / 2)
#define CT_CHUNK 128                        // tasks staged in LDS at a time
#define SYNC_FLAGS OVHIP_FLOW_SYNC_WORDS     // sync[0] = abort code; flags from word 16
#ifndef OVHIP_FLOW_PRIO
#define OVHIP_FLOW_PRIO 3
#endif
#ifndef FLOW_POLL_GAP
#define FLOW_NAP_HALF_LEVEL 24             // s_sleep units (64 clocks): two of them ~1.3 us, a bit more than half the best hop of the chain
#ifndef FLOW_NAP_MAX_LEVELS
#define FLOW_NAP_MAX_LEVELS 12             // (0: no nap -- the comparison build of tools/ipic_traffic.sh)
#endif
#define FLOW_POLL_GAP 3                    // s_sleep units (64 clocks) between the two polls a waiting item keeps in flight
#endif
#define SPIN_LIMIT (1u << 17)              // polls before a workgroup gives up (>= 40 ms; a legitimate wait is a few ms): ovhip_job_wait then decodes the picture per level

struct CtuLds {
    uint16_t ty[CT_S][CT_YS], top_y[4 + 2 * CT_S + 4];
    uint16_t tc[2][CT_S / 2][CT_CS], top_c[2][4 + CT_S + 4];
    int16_t ry[CT_S][CT_S], rc[2][CT_S / 2][CT_S / 2];      // the CTU's stored residuals (a global round trip per level otherwise)
    ovhip_itask task[CT_CHUNK];
    IntraLds w[4];
    int sc_idx[4], sc_val[4], n_sc, abort;
};

typedef unsigned long long u64;
// OVHIP_CTU_PROBE (debug builds only, tools/debug/ctu_probe.py): per-CTU wall-clock stamps (100 MHz) of the phases
#ifdef OVHIP_CTU_PROBE
__device__ u64 *g_probe;
#define PROBE(k) do { if (tid == 0 && g_probe && blockIdx.x < 64 && (k) < 256) g_probe[blockIdx.x * 256 + (k)] = wall_clock64(); } while (0)
#else
#define PROBE(k) do { } while (0)
#endif
// the same switch stamps the phases of every item of the flow launch (tools/debug/flow_probe.py): 8 stamps per item
#ifdef OVHIP_CTU_PROBE
__device__ const uint32_t *g_probe_items;      // the picture's first item: launches of later chunks index the stamps from it
// stamps are kept in registers and written once, behind the item's last one: a store per stamp (with the two loads of its
// address) put ~0.2 us of probe into every phase it closed
#define FPROBE(k) do { fprobe_[k] = wall_clock64(); \
                       if ((k) == 7 && lane == 0 && g_probe) { u64 *o_ = g_probe + (size_t)(items + bid - g_probe_items) * 8; \
                                                                for (int q_ = 0; q_ < 8; ++q_) o_[q_] = fprobe_[q_]; } } while (0)
#define FPROBE_DECL u64 fprobe_[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }
#else
#define FPROBE(k) do { } while (0)
#define FPROBE_DECL do { } while (0)
#endif
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ ovhip_itask uniform_task(const ovhip_itask *p)
{
    union { ovhip_itask t; unsigned u[8]; } v;
    const unsigned *q = reinterpret_cast<const unsigned *>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v.u[i] = __builtin_amdgcn_readfirstlane(q[i]);
    return v.t;
}

// one (task, strip, plane) item by one wave; samples from / to the tile
__device__ void ctu_item(CtuLds &L, IntraLds &s, const ovhip_itask &t, int strip, int comp, int X0, int Y0,
                         const ovhip_lmcs_region *__restrict__ regs, const LmcsWnd &wnd, int16_t *__restrict__ scales, int log2_ctu, int lane)
{
    const int l2w = t.log2_w, w = 1 << l2w, h = 1 << t.log2_h, npx = w * h;
    const TileAcc ya = { &L.ty[0][0], L.top_y, CT_YS, X0, Y0 };
    if (t.kind == OVHIP_IT_REGION) {
        const int v = region_scale(ya, regs[t.c_scale], wnd, lane);
        if (lane == 0) { const int k = atomicAdd(&L.n_sc, 1) & 3; L.sc_idx[k] = t.c_scale; L.sc_val[k] = v; scales[t.c_scale] = (int16_t)v; }
        return;
    }
    const bool luma = t.kind == OVHIP_IT_LUMA;
    Strip st; st.p0 = strip * STRIP; st.p1 = min(npx, st.p0 + STRIP);
    const bool has_res = t.flags & (luma ? OVHIP_IF_RES_Y : (comp ? OVHIP_IF_RES_CR : OVHIP_IF_RES_CB));
    const bool res_only = t.kind == OVHIP_IT_RES_C;
    if (res_only && !has_res) return;
    const bool scaled = !luma && (t.flags & OVHIP_IF_RES_SCALE);
    int scale = 0;
    if (scaled) {
        if (t.flags & OVHIP_IF_SCALE_IDX) {
            int found = -1;
#pragma unroll
            for (int k = 0; k < 4; ++k) if (k < min(L.n_sc, 4) && L.sc_idx[k] == t.c_scale) found = L.sc_val[k];
            scale = found >= 0 ? found : scales[t.c_scale];          // not from this CTU: k_lmcs_scale wrote it before this launch
        } else scale = t.c_scale;
    }
    const int tx = (luma ? t.x - X0 : t.x - (X0 >> 1)) + 4, ty = luma ? t.y - Y0 : t.y - (Y0 >> 1);
    uint16_t *dst = luma ? &L.ty[ty][tx] : &L.tc[comp][ty][tx];
    const int16_t *rp = luma ? &L.ry[ty][tx - 4] : &L.rc[comp][ty][tx - 4];
    const int dstride = luma ? CT_YS : CT_CS, rstride = luma ? CT_S : CT_S / 2;
    const int res_mask = (t.flags & OVHIP_IF_ISP) ? t.isp_res_mask : 0xff, res_l2pb = (t.flags & OVHIP_IF_ISP) ? t.isp_log2_pb : 6;
    const int ciip_wt = res_only ? 0 : t.ciip_wt;
    if (!res_only) {
        if (luma) {
            if (t.flags & OVHIP_IF_ISP) fetch_refs_isp(s, ya, t, lane);
            else fetch_refs(s, ya, t.x, t.y, w, h, 4, t.flags & OVHIP_IF_CORNER, t.avl_abv, t.avl_lft, (t.flags & OVHIP_IF_MIP) ? 0 : t.mrl_idx, lane);
            wave_sync();
            if (t.flags & OVHIP_IF_MIP) pred_mip(s, t, st, lane);
            else pred_regular(s, t, true, st, lane);
        } else {
            const TileAcc ca = { &L.tc[comp][0][0], L.top_c[comp], CT_CS, X0 >> 1, Y0 >> 1 };
            if (t.mode >= 67) pred_cclm<NPL>(s, ya, ca, t, log2_ctu, st, lane);
            else {
                fetch_refs(s, ca, t.x, t.y, w, h, 2, t.flags & OVHIP_IF_CORNER, t.avl_abv, t.avl_lft, 0, lane);
                wave_sync();
                pred_regular(s, t, false, st, lane);
            }
        }
        wave_sync();
    }
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int p = st.p0 + lane + 64 * i, x = p & (w - 1), y = p >> l2w;
        if (p >= st.p1) break;
        uint16_t *d = dst + y * dstride + x;
        int v = res_only ? (int)*d : (int)s.pred[p - st.p0];
        if (ciip_wt) v = (v * ciip_wt + (int)*d * (4 - ciip_wt) + 2) >> 2;
        if (has_res && ((res_mask >> (x >> res_l2pb)) & 1)) { const int r = rp[y * rstride + x]; v = ov_clip_bd(v + (scaled ? res_scale(r, scale) : r)); }
        *d = (uint16_t)v;
    }
}

__global__ __launch_bounds__(256) void k_intra_ctu(ovhip_pic pic, ovhip_pic res, const ovhip_itask *__restrict__ tasks,
                                                   const ovhip_ictu *__restrict__ ctus, const ovhip_lmcs_region *__restrict__ regs, LmcsWnd wnd,
                                                   int16_t *__restrict__ scales, int log2_ctu, unsigned *sync, unsigned epoch, int ncx,
                                                   unsigned *abort_mirror)
{
    __shared__ CtuLds L;
    const ovhip_ictu c = ctus[blockIdx.x];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int S = 1 << log2_ctu, X0 = c.cx << log2_ctu, Y0 = c.cy << log2_ctu;
    const int cw = min(S, pic.w - X0), ch = min(S, pic.h - Y0);
    const int wc = pic.w >> 1, hc = pic.h >> 1, Sc = S >> 1, X0c = X0 >> 1, Y0c = Y0 >> 1, cwc = cw >> 1, chc = ch >> 1;
    unsigned *flags = sync + SYNC_FLAGS;
    if (tid == 0) { L.n_sc = 0; L.abort = 0; }
    load_tables(L.w[wave], lane);
    __syncthreads();

    // ---- 1. the neighbours this CTU reads must have published their tiles ----
    if (c.deps) {
        if (wave == 0) {
            bool ok = true;
            if (lane < 4 && ((c.deps >> lane) & 1)) {
                const int nx = c.cx + (lane == 3 ? 1 : (lane == 2 ? 0 : -1)), ny = c.cy - (lane != 0);
                unsigned *f = flags + ny * ncx + nx;
                unsigned spins = 0;
                while (__hip_atomic_load(f, RLX_AGENT) != epoch) {
                    if (++spins > (SPIN_LIMIT << 4) || __hip_atomic_load(sync, RLX_AGENT) != 0) { ok = false; break; }
                    __builtin_amdgcn_s_sleep(8);
                }
            }
            if (!__all(ok)) {
                if (lane == 0) {
                    __hip_atomic_store(sync, 1u + blockIdx.x, RLX_AGENT);
                    if (abort_mirror) __hip_atomic_store(abort_mirror, 1u + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    L.abort = 1;
                }
            }
            else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (L.abort) return;
    }

    PROBE(0);
    // ---- 2. the CTU and its borders into LDS (8-byte granules; widths are multiples of 8) ----
    {
        const int gx0 = X0 ? -1 : 0;                                   // first granule: the left border
        const int ng = (cw >> 2) - gx0;
        for (int i = tid; i < ch * ng; i += 256) {
            const int r = i / ng, g = i - r * ng + gx0;
            *reinterpret_cast<u64 *>(&L.ty[r][4 + 4 * g]) = *reinterpret_cast<const u64 *>(pic.y + (size_t)(Y0 + r) * pic.stride_y + X0 + 4 * g);
        }
        const int ngc = (cwc >> 2) - gx0;
        for (int i = tid; i < 2 * chc * ngc; i += 256) {
            const int pl = i >= chc * ngc, k = i - pl * chc * ngc, r = k / ngc, g = k - r * ngc + gx0;
            const uint16_t *src = pl ? pic.cr : pic.cb;
            *reinterpret_cast<u64 *>(&L.tc[pl][r][4 + 4 * g]) = *reinterpret_cast<const u64 *>(src + (size_t)(Y0c + r) * pic.stride_c + X0c + 4 * g);
        }
        const int nr = cw >> 2, nrc = cwc >> 2;                        // residuals: written by the launches before this one
        for (int i = tid; i < ch * nr; i += 256) {
            const int r = i / nr, g = i - r * nr;
            *reinterpret_cast<u64 *>(&L.ry[r][4 * g]) = *reinterpret_cast<const u64 *>(res.y + (size_t)(Y0 + r) * res.stride_y + X0 + 4 * g);
        }
        for (int i = tid; i < 2 * chc * nrc; i += 256) {
            const int pl = i >= chc * nrc, k = i - pl * chc * nrc, r = k / nrc, g = k - r * nrc;
            const uint16_t *src = pl ? res.cr : res.cb;
            *reinterpret_cast<u64 *>(&L.rc[pl][r][4 * g]) = *reinterpret_cast<const u64 *>(src + (size_t)(Y0c + r) * res.stride_c + X0c + 4 * g);
        }
        if (Y0) {
            const int nt = (min(pic.w, X0 + cw + S) - X0) / 4 - gx0, ntc = (min(wc, X0c + cwc + Sc) - X0c) / 4 - gx0;
            for (int i = tid; i < nt + 2 * ntc; i += 256) {
                if (i < nt) {
                    const int g = i + gx0;
                    *reinterpret_cast<u64 *>(&L.top_y[4 + 4 * g]) = *reinterpret_cast<const u64 *>(pic.y + (size_t)(Y0 - 1) * pic.stride_y + X0 + 4 * g);
                } else {
                    const int k = i - nt, pl = k >= ntc, g = k - pl * ntc + gx0;
                    const uint16_t *src = pl ? pic.cr : pic.cb;
                    *reinterpret_cast<u64 *>(&L.top_c[pl][4 + 4 * g]) = *reinterpret_cast<const u64 *>(src + (size_t)(Y0c - 1) * pic.stride_c + X0c + 4 * g);
                }
            }
        }
    }

    PROBE(1);
    // ---- 3. the CTU's tasks: runs of equal level, the items of a run spread over the four waves ----
    int n_run = 0;
    for (unsigned base = 0; base < c.n; base += CT_CHUNK) {
        const int nchunk = (int)min((unsigned)CT_CHUNK, c.n - base);
        __syncthreads();
        if (tid < 2 * nchunk) reinterpret_cast<uint4 *>(L.task)[tid] = reinterpret_cast<const uint4 *>(tasks + c.first + base)[tid];
        __syncthreads();
        int i = 0;
        while (i < nchunk) {
            const int lvl = __builtin_amdgcn_readfirstlane((int)L.task[i].level);
            int cnt = 0, j = i;
            for (; j < nchunk; ++j) {
                const ovhip_itask t = uniform_task(&L.task[j]);
                if (t.level != lvl) break;
                const int npx = 1 << (t.log2_w + t.log2_h);
                const int strips = t.kind == OVHIP_IT_REGION ? 1 : (npx + STRIP - 1) / STRIP, comps = (t.kind == OVHIP_IT_CHROMA || t.kind == OVHIP_IT_RES_C) ? 2 : 1;
                for (int st = 0; st < strips; ++st)
                    for (int cp = 0; cp < comps; ++cp)
                        if ((cnt++ & 3) == wave) ctu_item(L, L.w[wave], t, st, cp, X0, Y0, regs, wnd, scales, log2_ctu, lane);
            }
            __syncthreads();
            i = j;
            PROBE(4 + n_run); ++n_run;
        }
    }
    PROBE(2);

    // ---- 4. publish: write-through stores, every wave drains, one flag ----
    {
        const int ng = cw >> 2;
        for (int i = tid; i < ch * ng; i += 256) {
            const int r = i / ng, g = i - r * ng;
            __hip_atomic_store(reinterpret_cast<u64 *>(pic.y + (size_t)(Y0 + r) * pic.stride_y + X0 + 4 * g), *reinterpret_cast<const u64 *>(&L.ty[r][4 + 4 * g]), RLX_AGENT);
        }
        const int ngc = cwc >> 2;
        for (int i = tid; i < 2 * chc * ngc; i += 256) {
            const int pl = i >= chc * ngc, k = i - pl * chc * ngc, r = k / ngc, g = k - r * ngc;
            uint16_t *dstp = pl ? pic.cr : pic.cb;
            __hip_atomic_store(reinterpret_cast<u64 *>(dstp + (size_t)(Y0c + r) * pic.stride_c + X0c + 4 * g), *reinterpret_cast<const u64 *>(&L.tc[pl][r][4 + 4 * g]), RLX_AGENT);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flags + c.cy * ncx + c.cx, epoch, RLX_AGENT);
    PROBE(3);
    (void)hc; (void)n_run;
}


// ------------------------------------------------------------------------------------------------------------------------
// The ordered pass as ONE launch with per-unit dependency flags ("flow").
//
// Items = (task, strip, plane) in LEVEL order, one 64-lane workgroup each.  Every 4x4-luma unit of every plane has a state word:
// 2 * epoch = "an ordered task of this picture will write it" (set by k_intra_flow_prepare before this launch),
// 2 * epoch + 1 = "written".  An item polls the units its reference arms cover (all of them at once, one per lane), reads the
// samples with agent-scope loads (bypass L1; the producer stored write-through), predicts, stores write-through, drains, and
// marks its own units.  No launch boundary per level, no cold fetch of the task record on the critical path (the workgroup
// has it long before its inputs are ready).  Forward progress: an item waits only for items of a LOWER level = lower index;
// workgroups start in index order on this hardware, nothing promises it -- the wait is bounded and aborts the launch.
// (bit 15 of a luma sample: FLOW_TAG below; every reader of the picture inside the flow launch drops it)
struct AgentAcc {
    const uint16_t *p; int stride;
    __device__ __forceinline__ int ld(int x, int y) const { return __hip_atomic_load(p + y * stride + x, RLX_AGENT) & 0x7fff; }
};
// (FlowState and flow_prepare_task: flow_state.hip.h)
// Luma hand-over inside the flow launch: DATA-TAGGED samples.  A luma item stores its samples with bit 15 set (10-bit samples leave
// it free) and a luma item that needs them polls the reference samples themselves until the bit is there: the flag round trips
// (drain the stores, mark the units, the consumer's poll, THEN its loads of the data) shrink to the one trip of the data.  Every
// luma sample the pass starts with is clean (written by this picture's motion compensation / transform launches or by a previous
// picture's filters), so the bit alone says "an ordered task of this picture has written this".  Whether a reference sample will
// carry it is the unit's state word (set by k_intra_flow_prepare, never polled).  The units are still marked written afterwards:
// chroma items (CCLM), scale regions and ISP partitions keep waiting on the words.  The bit is gone after the ordered pass: the
// inverse luma mapping indexes its table with the low 10 bits, pictures without LMCS get k_flow_untag.
#define FLOW_TAG 0x8000u

// reference arms of a regular luma block out of tagged samples (the coordinates exactly as fetch_refs picks them).  false: gave up.
// (luma: unit 4, state words fs.y; a chroma plane: unit 2, its own words -- a 2x2 chroma unit is the 4x4 luma unit)
__device__ __forceinline__ bool fetch_refs_tagged(IntraLds &s, const uint16_t *py, int stride, const unsigned *state, int w4, unsigned epoch, unsigned *sync,
                                                  int unit, int x0, int y0, int w, int h, bool corner, int avl_abv, int avl_lft, int mrl, int lane)
{
    const int l2u = unit == 4 ? 2 : 1;
    const int na = 2 * w + mrl + 1, nl = 2 * h + mrl + 1;
    const int cx = x0 - 1 - mrl, cy = y0 - 1 - mrl;
    const bool none = !corner && !avl_abv && !avl_lft;
    const int la = min(mrl + avl_abv * unit, na - 1), ll = min(mrl + avl_lft * unit, nl - 1);
    const int ax1 = cx + mrl + 1, ay1 = cy, lx1 = cx, ly1 = cy + mrl + 1;
    int v[6];
    bool ok = true;
    if (none) {
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] = 1 << (OV_BD - 1);
    } else {
        const bool fa0 = (mrl == 0 && avl_abv && avl_lft) || !avl_lft;
        const int cax = corner ? 0 : (fa0 ? ax1 : lx1), cay = fa0 ? ay1 : ly1;
        const int eax = avl_abv ? cx + la : (corner ? cx + mrl : lx1), eay = (avl_abv || corner) ? cy : ly1;
        const int clx = avl_lft ? lx1 : ax1, cly = avl_lft ? ly1 : ay1;
        const int elx = (avl_lft || corner) ? cx : ax1, ely = avl_lft ? cy + ll : (corner ? cy + mrl : ay1);
        const uint16_t *src[6];
        bool expect[6];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int k = lane + 64 * i;
            const int ka = min(k, na - 1), kl = min(k, nl - 1);
            const bool a_corner = ka <= mrl, a_avail = ((ka - mrl - 1) >> l2u) < avl_abv;
            const int ax = a_corner ? (corner ? cx + ka : cax) : (a_avail ? cx + ka : eax);
            const int ay = a_corner ? (corner ? cy : cay) : (a_avail ? cy : eay);
            const bool l_corner = kl <= mrl, l_avail = ((kl - mrl - 1) >> l2u) < avl_lft;
            const int lx = l_corner ? (corner ? cx : clx) : (l_avail ? cx : elx);
            const int ly = l_corner ? (corner ? cy + kl : cly) : (l_avail ? cy + kl : ely);
            src[2 * i] = py + ay * stride + ax; src[2 * i + 1] = py + ly * stride + lx;
            // will an ordered task of this picture write the sample?  (2 * epoch: it will; 2 * epoch + 1: it has)
            expect[2 * i] = (__hip_atomic_load(state + (ay >> l2u) * w4 + (ax >> l2u), RLX_AGENT) >> 1) == epoch;
            expect[2 * i + 1] = (__hip_atomic_load(state + (ly >> l2u) * w4 + (lx >> l2u), RLX_AGENT) >> 1) == epoch;
        }
        // Two polls in flight, half a round trip apart (loads return in order: waiting for the older one leaves the younger in flight):
        // a poll completes every ~0.3 us instead of every round trip + nap (~0.65 us), so a producer's store is seen ~0.15 us after it
        // lands instead of ~0.33 -- on every hop of the chain.  Every poll re-reads all six samples (a tagged sample stays tagged).
        auto missing = [&](const int *q) { bool m = false;
#pragma unroll
                                           for (int i = 0; i < 6; ++i) m |= expect[i] && !(q[i] & FLOW_TAG);
                                           return __any(m); };
        int va[6], vb[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) va[i] = __hip_atomic_load(src[i], RLX_AGENT);
        __builtin_amdgcn_s_sleep(FLOW_POLL_GAP);
#pragma unroll
        for (int i = 0; i < 6; ++i) vb[i] = __hip_atomic_load(src[i], RLX_AGENT);
        unsigned spins = 0;
        // (a ring of N polls written as a loop over arrays measured slower than this -- 2.55 us per level with two, 2.81 with three, 3.09
        // with four against 2.32: every generation still in flight when the samples arrive is waited for by whatever needs its registers)
        // (a waiting item polls at the lowest issue priority: the polls of the ~1000 workers that are levels ahead of the front must not
        //  take issue slots from the kernels of the other pictures -- the launch runs at OVHIP_FLOW_PRIO for what follows the arrival;
        //  on the stream of bench.py the difference is inside the noise: 3243-3462 without, 3309-3406 with)
        bool lowered = false;
        for (;;) {
            if (!missing(va)) {
#pragma unroll
                for (int i = 0; i < 6; ++i) v[i] = va[i];
                break;
            }
            if (!lowered) { __builtin_amdgcn_s_setprio(0); lowered = true; }
            const unsigned stop = __hip_atomic_load(sync, RLX_AGENT);
#pragma unroll
            for (int i = 0; i < 6; ++i) va[i] = __hip_atomic_load(src[i], RLX_AGENT);              // (behind the poll in vb)
            if (!missing(vb)) {
#pragma unroll
                for (int i = 0; i < 6; ++i) v[i] = vb[i];                                          // (the poll in va is left to arrive)
                break;
            }
            if (__any(++spins > SPIN_LIMIT || stop != 0)) { ok = false; break; }
#pragma unroll
            for (int i = 0; i < 6; ++i) vb[i] = __hip_atomic_load(src[i], RLX_AGENT);              // (behind the poll in va)
        }
        if (lowered) __builtin_amdgcn_s_setprio(OVHIP_FLOW_PRIO);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int k = lane + 64 * i;
        if (k < na + 24) s.abv[IR_NEG + k] = (uint16_t)(v[2 * i] & 0x7fff);
        if (k < nl + 24) s.lft[IR_NEG + k] = (uint16_t)(v[2 * i + 1] & 0x7fff);
    }
    return ok;
}
#define FLOW_MAX_FP 448
#ifndef FSTRIP
#define FSTRIP 256
#endif
// FSTRIP: samples per item: one wave predicts a 1024-sample strip in ~2 us, the critical path of a hop
#define FNPL   (FSTRIP / 64)
#define FJ8    ((FSTRIP + 511) / 512)        // runs of 8 / of 4 samples per lane
#define FJ4    ((FSTRIP + 255) / 256)
#define FREG   (8 * FJ8 > FNPL ? 8 * FJ8 : FNPL)

// 16 lanes per task, 16 tasks per workgroup (one 64-lane workgroup per task was 16k tiny workgroups for a B picture: 10 us)
__global__ __launch_bounds__(256) void k_intra_flow_prepare(const ovhip_itask *__restrict__ tasks, uint32_t n, FlowState fs, unsigned epoch)
{
    const uint32_t ti = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (ti < n) flow_prepare_task(tasks[ti], fs, epoch, threadIdx.x & 15);
}

typedef uint32_t flow_u4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t flow_u2 __attribute__((ext_vector_type(2), aligned(4)));
// The blocks the flow launch wrote, without their hand-over bit: the chroma blocks always, the luma blocks when nothing else drops it
// (pictures without LMCS)
__global__ __launch_bounds__(256) void k_flow_untag(ovhip_pic pic, const ovhip_itask *__restrict__ tasks, uint32_t n, int with_luma)
{
    // 16 lanes per task, 16 tasks per workgroup; a lane clears runs of 4 samples (8 bytes) -- blocks narrower than 4: of 2
    const uint32_t ti = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (ti >= n) return;
    const ovhip_itask t = tasks[ti];
    if (t.kind == OVHIP_IT_REGION || (t.kind == OVHIP_IT_LUMA && !with_luma)) return;
    const int lane = threadIdx.x & 15;
    const int l2w = t.log2_w, w = 1 << l2w, npx = w << t.log2_h;
    const bool luma = t.kind == OVHIP_IT_LUMA;
    const int stride = luma ? pic.stride_y : pic.stride_c;
    for (int c = 0; c < (luma ? 1 : 2); ++c) {
        if (!luma && t.kind == OVHIP_IT_RES_C && !(t.flags & (c ? OVHIP_IF_RES_CR : OVHIP_IF_RES_CB))) continue;
        uint16_t *dst = (luma ? pic.y : (c ? pic.cr : pic.cb)) + t.y * stride + t.x;
        if (w >= 4) {
            for (int p = 4 * lane; p < npx; p += 64) {
                flow_u2 *q = reinterpret_cast<flow_u2 *>(dst + (p >> l2w) * stride + (p & (w - 1)));
                flow_u2 v = *q; v[0] &= 0x03ff03ffu; v[1] &= 0x03ff03ffu; *q = v;
            }
        } else if (w == 2) {
            for (int p = 2 * lane; p < npx; p += 32) {
                uint32_t *q = reinterpret_cast<uint32_t *>(dst + (p >> l2w) * stride + (p & (w - 1)));
                *q &= 0x03ff03ffu;
            }
        } else {
            for (int p = lane; p < npx; p += 16) { uint16_t *q = dst + (p >> l2w) * stride + (p & (w - 1)); *q = *q & 0x3ff; }
        }
    }
}

// (fp: the state words an item polls, as word offsets from the state block: 7 KB of LDS per workgroup instead of 8.9.  The 109 VGPRs
//  hold the kernel at 16 workgroups per compute unit; amdgpu_waves_per_eu(5, 8) gives 96 VGPRs = 20 per compute unit = room for
//  1280 workers per launch, measured: the B picture's pass 157 -> 142 us alone, the I picture's 2.32 -> 2.44 us per level, the stream
//  the same within its noise -- not taken)
struct FlowLds { IntraLds s; uint32_t fp[FLOW_MAX_FP]; int abort; };

__global__ __launch_bounds__(64) void k_intra_flow(ovhip_pic pic, ovhip_pic res, const ovhip_itask *__restrict__ tasks, const uint32_t *__restrict__ items,
                                                   uint32_t n_items, const ovhip_lmcs_region *__restrict__ regs, LmcsWnd wnd, int16_t *scales,
                                                   int log2_ctu, FlowState fs, unsigned epoch, unsigned *sync, unsigned *abort_mirror, int nap)
{
    __shared__ FlowLds L;
    IntraLds &s = L.s;
    if (blockIdx.x >= n_items) return;
    // An item is one link of a dependency chain that a whole picture (and, for an I picture, everything decoded after it) waits for,
    // and it shares its SIMD with the throughput kernels of the other pictures in flight: highest issue priority (an I picture's pass
    // beside 16 B pictures ran at 9 us per level instead of 2.8 alone)
    __builtin_amdgcn_s_setprio(OVHIP_FLOW_PRIO);
    const int lane = threadIdx.x;
    load_tables(s, lane);
    // The workgroups of the launch are WORKERS: worker b takes the items b, b + W, b + 2 W, ... (W = gridDim.x) in ascending order.
    // An item waits only for items of lower index, so the lowest unfinished item always belongs to a worker that is working on it or
    // will be once it is resident: no deadlock as long as the W workers of a launch all become resident, which they do when the
    // launches in flight ask for fewer wave slots than the device has -- W is the bound on this launch's pollers (r2 launched one
    // workgroup per item: the device filled with the pollers of levels far ahead, which starved the kernels of the other pictures and
    // each other; r3 first capped them with LDS they did not use, which took that LDS from everybody else).
    int prev_level = 0;                                     // level of this worker's previous item (0: none yet)
    for (uint32_t bid = blockIdx.x; bid < n_items; bid += gridDim.x) {
    if (bid != blockIdx.x) wave_sync();                    // the tiles of the item before are dead
    FPROBE_DECL;
    const uint32_t item = items[bid];
    const ovhip_itask t = tasks[item & 0xffffff];
    const int strip = (item >> 24) & 0x1f, comp = (item >> 29) & 1;
    const int l2w = t.log2_w, w = 1 << l2w, h = 1 << t.log2_h, npx = w * h;
    const bool luma = t.kind == OVHIP_IT_LUMA, region = t.kind == OVHIP_IT_REGION, res_only = t.kind == OVHIP_IT_RES_C;
    const bool lm = t.kind == OVHIP_IT_CHROMA && t.mode >= 67 && !(t.flags & OVHIP_IF_BDPCM);
    const unsigned pending = 2 * epoch;
    const int w4 = fs.w4;
    Strip st; st.p0 = strip * FSTRIP; st.p1 = min(npx, st.p0 + FSTRIP);
    const bool has_res = t.flags & (luma ? OVHIP_IF_RES_Y : (comp ? OVHIP_IF_RES_CR : OVHIP_IF_RES_CB));
    if (res_only && !has_res) continue;

    // ---- what does not depend on the other ordered tasks (residual, inter prediction): requested before the wait ----
    uint16_t *pl = luma ? pic.y : (comp ? pic.cr : pic.cb);
    const int dstride = luma ? pic.stride_y : pic.stride_c, rstride = luma ? res.stride_y : res.stride_c;
    uint16_t *dst = pl + t.y * dstride + t.x;
    const int16_t *rp = reinterpret_cast<const int16_t *>(luma ? res.y : (comp ? res.cr : res.cb)) + t.y * rstride + t.x;
    const int ciip_wt = res_only ? 0 : t.ciip_wt;
    const bool need_d = ciip_wt || res_only;
    const int res_mask = (t.flags & OVHIP_IF_ISP) ? t.isp_res_mask : 0xff, res_l2pb = (t.flags & OVHIP_IF_ISP) ? t.isp_log2_pb : 6;
    // A lane owns runs of G consecutive samples of a row (G = 8, 4 or 1 by block width): residual and output move as 16- / 8-byte
    // accesses -- a write-through store is one fabric write whatever its size, and 2-byte ones cost 12x the time per byte
    const int l2g = w >= 8 ? 3 : (w == 4 ? 2 : 0);
    auto p_of = [&](int i) { return st.p0 + (((lane + 64 * (i >> l2g)) << l2g) | (i & ((1 << l2g) - 1))); };
    // (raw vectors: unpacked in the epilogue, so that nothing waits for these loads before the dependency wait has ended)
    flow_u4 rq8[FJ8], dq8[FJ8];
    flow_u2 rq4[FJ4], dq4[FJ4];
    int rv1[FNPL], dv1[FNPL];
#pragma unroll
    for (int j = 0; j < FJ8; ++j) { rq8[j] = 0; dq8[j] = 0; }
#pragma unroll
    for (int j = 0; j < FJ4; ++j) { rq4[j] = 0; dq4[j] = 0; }
#pragma unroll
    for (int i = 0; i < FNPL; ++i) { rv1[i] = 0; dv1[i] = 0; }
    if (!region && (has_res || need_d)) {
        if (l2g == 3) {
#pragma unroll
            for (int j = 0; j < FJ8; ++j) {
                const int p = min(p_of(8 * j), st.p1 - 8), x = p & (w - 1), y = p >> l2w;       // lanes past the strip re-read its last run
                if (has_res) rq8[j] = *reinterpret_cast<const flow_u4 *>(rp + y * rstride + x);
                if (need_d) dq8[j] = *reinterpret_cast<const flow_u4 *>(dst + y * dstride + x);
            }
        } else if (l2g == 2) {
#pragma unroll
            for (int j = 0; j < FJ4; ++j) {
                const int p = min(p_of(4 * j), st.p1 - 4), x = p & (w - 1), y = p >> l2w;
                if (has_res) rq4[j] = *reinterpret_cast<const flow_u2 *>(rp + y * rstride + x);
                if (need_d) dq4[j] = *reinterpret_cast<const flow_u2 *>(dst + y * dstride + x);
            }
        } else {
#pragma unroll
            for (int i = 0; i < FNPL; ++i) {
                const int p = min(p_of(i), st.p1 - 1), x = p & (w - 1), y = p >> l2w;
                if (has_res) rv1[i] = rp[y * rstride + x];                       // residuals: the launches before
                if (need_d) dv1[i] = dst[y * dstride + x];                       // inter prediction: likewise
            }
        }
    }
    // ---- nap before the first poll (VERDICT r3 #5b).  An item of level L reads what an item of level L - 1 writes, and that one what
    // level L - 2 wrote: when this worker finished an item of level P just now, the inputs of its next item cannot exist before
    // L - P - 1 more links of the chain have run, ~2.3 us each at best.  In an I picture a worker's next item is W items
    // = ~10 levels ahead: it used to poll through all of that (92 of the pass's 101 MB of traffic per launch were polls).  Napping
    // ~1.3 us per level of the gap (capped) leaves the last stretch to the polls; B pictures' wide levels give gaps of 0 or 1 ----
    {
        const int gap = min((int)t.level - prev_level - 1, FLOW_NAP_MAX_LEVELS);
        prev_level = (int)t.level;
        if (gap > 0) {
            __builtin_amdgcn_s_setprio(0);
            for (int k = 0; k < 2 * gap; ++k) __builtin_amdgcn_s_sleep(FLOW_NAP_HALF_LEVEL);
            __builtin_amdgcn_s_setprio(OVHIP_FLOW_PRIO);
        }
    }
    // ---- what this item reads: unit state words, all polled at once ----
    FPROBE(0);
    int nfp = 0;
    auto add_run = [&](unsigned *base, int ux, int uy, int count, int dx, int dy) {
        // count units from (ux, uy) in steps of (dx, dy); clipped to the table (count is wave-uniform)
        count = min(count, FLOW_MAX_FP - nfp);
        for (int i = lane; i < count; i += 64) L.fp[nfp + i] = (uint32_t)(base - sync) + (uint32_t)((uy + i * dy) * w4 + ux + i * dx);
        nfp += max(count, 0);
    };
    if (region) {
        const ovhip_lmcs_region g = regs[t.c_scale];
        if (g.n_abv) add_run(fs.y, g.x >> 2, (g.y >> 2) - 1, g.n_abv, 1, 0);
        if (g.n_lft) add_run(fs.y, (g.x >> 2) - 1, g.y >> 2, g.n_lft, 0, 1);
    } else if (luma) {
        // (regular luma blocks wait on the tagged reference samples themselves, fetch_refs_tagged below: nothing to poll here)
        if (t.flags & OVHIP_IF_ISP) {
            // the above arm lies in row t.y - 1: unit row uya -- the row of units above this block, or, for a partition less than a unit
            // high that does not start a unit row (8x2 / 16x1 ... partitions: rows 2-3 / 1, 2, 3 of the unit), the block's OWN unit row
            const int ux = t.x >> 2, uya = (t.y - 1) >> 2, uxa = (t.x - t.isp_off_x) >> 2, uyl = (t.y - t.isp_off_y) >> 2;
            if (t.flags & OVHIP_IF_CORNER) add_run(fs.y, uxa - 1, uya, 1, 1, 0);
            if (t.flags & OVHIP_IF_CORNER_L) add_run(fs.y, ux - 1, uyl - 1, 1, 1, 0);
            if (t.y & 3) {
                // inside the coding unit's columns that row is the partition before this one, in the same units as this block: their
                // word turns "written" only with the unit's LAST partition (below), so the samples themselves are waited for (thin_row);
                // right of the coding unit the arm runs through units of its own
                const int ncu = (1 << t.isp_log2_cb_w) >> 2;
                if ((int)t.avl_abv > ncu) add_run(fs.y, uxa + ncu, uya, (int)t.avl_abv - ncu, 1, 0);
            } else add_run(fs.y, uxa, uya, t.avl_abv, 1, 0);
            add_run(fs.y, ux - 1, uyl, t.avl_lft, 0, 1);
        }
    } else {
        unsigned *fc = fs.c[comp];
        const int ux = t.x >> 1, uy = t.y >> 1, nux = max(1, w >> 1), nuy = max(1, h >> 1);
        if (!res_only) {
            if (lm) {
                // co-located luma, its left / above margin, and the chroma + luma neighbours the parameters are derived from
                const int na = t.mode == 69 ? t.avl_abv : (t.avl_abv ? nux : 0), nl = t.mode == 68 ? t.avl_lft : (t.avl_lft ? nuy : 0);
                for (int r = 0; r < nuy && nfp < FLOW_MAX_FP; ++r) add_run(fs.y, ux, uy + r, nux, 1, 0);
                if (t.avl_lft) { add_run(fs.y, ux - 1, uy - (t.avl_abv ? 1 : 0), max(nl, nuy) + (t.avl_abv ? 1 : 0), 0, 1); add_run(fc, ux - 1, uy, nl, 0, 1); }
                if (t.avl_abv) { add_run(fs.y, ux, uy - 1, max(na, nux), 1, 0); add_run(fc, ux, uy - 1, na, 1, 0); }
            }      // (the other chroma modes wait on their tagged reference samples, like luma)
        }
        if ((t.flags & OVHIP_IF_RES_SCALE) && (t.flags & OVHIP_IF_SCALE_IDX)) add_run(fs.reg, t.c_scale, 0, 1, 1, 0);
    }
    wave_sync();
    FPROBE(1);
    {
        bool ok = true;
        for (int i = lane; i < nfp; i += 64) {
            unsigned *f = sync + L.fp[i];
            unsigned spins = 0;
            unsigned cur = __hip_atomic_load(f, RLX_AGENT);
            while (cur == pending) {
                if (++spins > SPIN_LIMIT) { ok = false; break; }
                if (nap == 0) __builtin_amdgcn_s_sleep(4);
                else if (nap == 1) __builtin_amdgcn_s_sleep(16);
                else if (nap == 2) __builtin_amdgcn_s_sleep(64);
                else { if (spins < 8) __builtin_amdgcn_s_sleep(4); else if (spins < 32) __builtin_amdgcn_s_sleep(16); else __builtin_amdgcn_s_sleep(64); }
                // (the abort word and the state word in ONE round trip)
                const unsigned stop = __hip_atomic_load(sync, RLX_AGENT);
                cur = __hip_atomic_load(f, RLX_AGENT);
                if (stop != 0) { ok = false; break; }
            }
        }
        if (!__all(ok)) {
            if (lane == 0) {
                __hip_atomic_store(sync, 1u + blockIdx.x, RLX_AGENT);
                if (abort_mirror) __hip_atomic_store(abort_mirror, 1u + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            return;
        }
    }
    FPROBE(2);

    const AgentAcc ya = { pic.y, pic.stride_y };
    if (region) {
        const int v = region_scale(ya, regs[t.c_scale], wnd, lane);
        if (lane == 0) {
            __hip_atomic_store(scales + t.c_scale, (int16_t)v, RLX_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(fs.reg + t.c_scale, pending + 1, RLX_AGENT);
        }
        continue;
    }
    const bool scaled = !luma && (t.flags & OVHIP_IF_RES_SCALE);
    // (the derived scale is requested here and first looked at in the epilogue: no wait in front of the reference fetch)
    const bool scale_idx = scaled && (t.flags & OVHIP_IF_SCALE_IDX);
    int scale_ld = 0;
    if (scale_idx) scale_ld = (int)__hip_atomic_load(scales + t.c_scale, RLX_AGENT);
    if (!res_only) {
        if (luma) {
            if ((t.flags & OVHIP_IF_ISP) && (t.y & 3)) {
                // thin_row: the last row of the partition above (same coding unit, same 4x4 units as this block) arrives TAGGED -- every
                // luma item stores its samples with FLOW_TAG -- and is polled itself, as fetch_refs_tagged polls a regular block's arms
                //  EVERY row the earlier partitions wrote into this partition's unit row, not only the last one: the partition
                // holding the unit's last row marks the unit "written" for the word readers (CCLM items, chroma-scale regions, ISP arms of
                // other coding units), who then read all four rows -- so all of them must have been observed by the marker
                const int cbw = 1 << t.isp_log2_cb_w;
                const int ry0 = (int)t.y & ~3, nry = (int)t.y - ry0;          // 1 .. 3 rows
                const uint16_t *row = pic.y + ry0 * pic.stride_y + (t.x - t.isp_off_x);
                unsigned spins = 0;
                bool ok = true;
                for (;;) {
                    unsigned v = FLOW_TAG;
                    for (int rr = 0; rr < nry; ++rr) v &= lane < cbw ? (unsigned)__hip_atomic_load(row + rr * pic.stride_y + lane, RLX_AGENT) : FLOW_TAG;
                    if (!__any(!(v & FLOW_TAG))) break;
                    const unsigned stop = __hip_atomic_load(sync, RLX_AGENT);
                    if (__any(++spins > SPIN_LIMIT || stop != 0)) { ok = false; break; }
                    __builtin_amdgcn_s_sleep(FLOW_POLL_GAP);
                }
                if (!ok) {
                    if (lane == 0) {
                        __hip_atomic_store(sync, 1u + blockIdx.x, RLX_AGENT);
                        if (abort_mirror) __hip_atomic_store(abort_mirror, 1u + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    return;
                }
            }
            if (t.flags & OVHIP_IF_ISP) fetch_refs_isp(s, ya, t, lane);
            else if (!fetch_refs_tagged(s, pic.y, pic.stride_y, fs.y, w4, epoch, sync, 4, t.x, t.y, w, h, t.flags & OVHIP_IF_CORNER, t.avl_abv, t.avl_lft,
                                        (t.flags & OVHIP_IF_MIP) ? 0 : t.mrl_idx, lane)) {
                if (lane == 0) {
                    __hip_atomic_store(sync, 1u + blockIdx.x, RLX_AGENT);
                    if (abort_mirror) __hip_atomic_store(abort_mirror, 1u + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                return;
            }
            wave_sync();
            FPROBE(3);
            if (t.flags & OVHIP_IF_MIP) pred_mip(s, t, st, lane);
            else pred_regular(s, t, true, st, lane);
        } else {
            const AgentAcc ca = { pl, pic.stride_c };
            if (lm) pred_cclm<FNPL>(s, ya, ca, t, log2_ctu, st, lane);
            else {
                if (!fetch_refs_tagged(s, pl, pic.stride_c, fs.c[comp], w4, epoch, sync, 2, t.x, t.y, w, h, t.flags & OVHIP_IF_CORNER, t.avl_abv, t.avl_lft, 0, lane)) {
                    if (lane == 0) {
                        __hip_atomic_store(sync, 1u + blockIdx.x, RLX_AGENT);
                        if (abort_mirror) __hip_atomic_store(abort_mirror, 1u + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    return;
                }
                wave_sync();
                pred_regular(s, t, false, st, lane);
            }
        }
        wave_sync();
    }
    FPROBE(4);
    // epilogue of a run of n samples (registers i0 .. i0 + n - 1): prediction out of LDS in ONE read, then blend / residual / clip
    // with the wave-uniform decisions outside the per-sample work (a chain of read-wait-branch per sample cost 0.8 us of the hop)
    const int scale = scale_idx ? scale_ld : (scaled ? (int)t.c_scale : 0);
    const uint32_t tag2 = FLOW_TAG | (FLOW_TAG << 16);                         // every sample leaves tagged (see FLOW_TAG)
    // (j: the run's index among the lane's runs; x: its first column, for the ISP partitions' residual mask)
    auto finish = [&](int j, int p, int x, int n, int *v) {
        int rv[8], dv[8];
        for (int e = 0; e < n; ++e) {
            const uint32_t rw = n == 8 ? rq8[j][e >> 1] : (n == 4 ? rq4[j][e >> 1] : (uint32_t)rv1[j]);
            const uint32_t dw = n == 8 ? dq8[j][e >> 1] : (n == 4 ? dq4[j][e >> 1] : (uint32_t)dv1[j]);
            const int r = n == 1 ? (int)(int16_t)rw : (int)(int16_t)(rw >> (16 * (e & 1)));
            rv[e] = ((res_mask >> ((x + e) >> res_l2pb)) & 1) ? r : 0;
            dv[e] = n == 1 ? (int)(dw & 0xffff) : (int)((dw >> (16 * (e & 1))) & 0xffff);
        }
        if (res_only) {
            for (int e = 0; e < n; ++e) v[e] = dv[e];
        } else if (n == 8) {
            const uint4 q = *reinterpret_cast<const uint4 *>(&s.pred[p - st.p0]);
            const uint32_t qq[4] = { q.x, q.y, q.z, q.w };
            for (int e = 0; e < 8; ++e) v[e] = (int)((qq[e >> 1] >> (16 * (e & 1))) & 0xffff);
        } else if (n == 4) {
            const uint2 q = *reinterpret_cast<const uint2 *>(&s.pred[p - st.p0]);
            const uint32_t qq[2] = { q.x, q.y };
            for (int e = 0; e < 4; ++e) v[e] = (int)((qq[e >> 1] >> (16 * (e & 1))) & 0xffff);
        } else v[0] = s.pred[p - st.p0];
        if (ciip_wt) for (int e = 0; e < n; ++e) v[e] = (v[e] * ciip_wt + dv[e] * (4 - ciip_wt) + 2) >> 2;
        if (has_res) {
            if (scaled) for (int e = 0; e < n; ++e) v[e] = ov_clip_bd(v[e] + res_scale(rv[e], scale));
            else        for (int e = 0; e < n; ++e) v[e] = ov_clip_bd(v[e] + rv[e]);
        }
    };
    // The plain case -- prediction (+ residual), no blend, no residual scale, every sample of the run carries the residual -- on
    // sample PAIRS: clip(pred + res) = min(max(sat16(pred + res), 0), 1023) is a saturating packed add and two packed min / max
    // (4 instructions per pair against ~20 unpacked ones: a lone wave issues an instruction every ~4.5 ns and this is the chain's tail)
    const bool plain = !ciip_wt && !res_only && !scaled && res_mask == 0xff;
    auto pk_clip_add = [&](uint32_t pred2, uint32_t res2) {
        uint32_t r;
        asm("v_pk_add_i16 %0, %1, %2 clamp\n\tv_pk_max_i16 %0, %0, 0\n\tv_pk_min_i16 %0, %0, %3" : "=&v"(r) : "v"(pred2), "v"(res2), "v"(0x03ff03ffu));
        return r;
    };
    if (l2g == 3) {
#pragma unroll
        for (int j = 0; j < FJ8; ++j) {
            const int p = p_of(8 * j), x = p & (w - 1), y = p >> l2w;
            if (p >= st.p1) break;
            flow_u4 q;
            if (plain) {
                const uint4 pq = *reinterpret_cast<const uint4 *>(&s.pred[p - st.p0]);
                q[0] = pq.x; q[1] = pq.y; q[2] = pq.z; q[3] = pq.w;
                if (has_res) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) q[e] = pk_clip_add(q[e], rq8[j][e]);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) q[e] |= tag2;
            } else {
                int v[8];
                finish(j, p, x, 8, v);
#pragma unroll
                for (int e = 0; e < 4; ++e) q[e] = ((uint32_t)v[2 * e] | ((uint32_t)v[2 * e + 1] << 16)) | tag2;
            }
            asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst + y * dstride + x), "v"(q) : "memory");      // write-through
        }
    } else if (l2g == 2) {
#pragma unroll
        for (int j = 0; j < FJ4; ++j) {
            const int p = p_of(4 * j), x = p & (w - 1), y = p >> l2w;
            if (p >= st.p1) break;
            flow_u2 q;
            if (plain) {
                const uint2 pq = *reinterpret_cast<const uint2 *>(&s.pred[p - st.p0]);
                q[0] = pq.x; q[1] = pq.y;
                if (has_res) { q[0] = pk_clip_add(q[0], rq4[j][0]); q[1] = pk_clip_add(q[1], rq4[j][1]); }
                q[0] |= tag2; q[1] |= tag2;
            } else {
                int v[4];
                finish(j, p, x, 4, v);
#pragma unroll
                for (int e = 0; e < 2; ++e) q[e] = ((uint32_t)v[2 * e] | ((uint32_t)v[2 * e + 1] << 16)) | tag2;
            }
            asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(dst + y * dstride + x), "v"(q) : "memory");
        }
    } else {
#pragma unroll
        for (int i = 0; i < FNPL; ++i) {
            const int p = p_of(i), x = p & (w - 1), y = p >> l2w;
            if (p >= st.p1) break;
            int v[1];
            finish(i, p, x, 1, v);
            __hip_atomic_store(dst + y * dstride + x, (uint16_t)(v[0] | (tag2 & 0xffff)), RLX_AGENT);
        }
    }
    FPROBE(5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FPROBE(6);
    // ---- this strip's units are written ----
    {
        const int sh = luma ? 2 : 1, y0s = st.p0 >> l2w, y1s = (st.p1 + w - 1) >> l2w;
        const int ux0 = t.x >> sh, uy0 = (t.y + y0s) >> sh, nx = max(1, w >> sh), ny = max(1, (y1s - y0s) >> sh);
        unsigned *fb = luma ? fs.y : fs.c[comp];
        const int l2nx = ilog2(nx);                                             // block widths are powers of two
        // (a luma partition less than a unit high that does not END its unit row leaves the words alone: the unit is written when the
        //  partition holding its last row is -- word readers: CCLM items, chroma-scale regions, ISP arms outside their coding unit)
        const bool unit_done = !(luma && h < 4 && ((t.y + h) & 3));
        for (int i = lane; unit_done && i < nx * ny; i += 64) __hip_atomic_store(fb + (uy0 + (i >> l2nx)) * w4 + ux0 + (i & (nx - 1)), pending + 1, RLX_AGENT);
    }
    FPROBE(7);
    }   // next item of this worker
}

} // namespace

// Launch geometry of a level from its tasks (HOST memory): bits 0-1 = log2 of the strips of the largest block (1024 samples
// each), bit 4 = some task covers both chroma planes.
extern "C" uint32_t ovhip_intra_level_geom(const ovhip_itask *tasks, size_t n)
{
    uint32_t l2s = 0, two = 0;
    for (size_t i = 0; tasks && i < n; ++i) {
        if (tasks[i].kind == OVHIP_IT_REGION) continue;
        const int l2 = tasks[i].log2_w + tasks[i].log2_h;
        if (l2 > 10 && (uint32_t)(l2 - 10) > l2s) l2s = (uint32_t)(l2 - 10);
        two |= tasks[i].kind != OVHIP_IT_LUMA;
    }
    return (l2s > 2 ? 2 : l2s) | (two << 4);
}

// One level of the ordered pass.  d_tasks: DEVICE, the n tasks of this level.  res: residual picture written by
// ovhip_itx_launch_classes_res.  d_regions / luts / d_scales: as ovhip_lmcs_scale_launch (may be NULL without LMCS chroma scaling).
// geom: ovhip_intra_level_geom() of the same tasks, or OVHIP_INTRA_GEOM_ANY.
extern "C" int ovhip_intra_level_launch(ovhip_ctx *ctx, const ovhip_pic *pic, const ovhip_pic *res, const ovhip_itask *d_tasks, uint32_t n,
                                        const ovhip_lmcs_region *d_regions, const ovhip_lmcs_luts *luts, int16_t *d_scales, int32_t log2_ctu_s,
                                        uint32_t geom)
{
    if (!ctx || !pic || !res) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    if (!n) return OVHIP_OK;
    if (!d_tasks || log2_ctu_s < 5 || log2_ctu_s > 7 || (geom & 3) == 3)
        return ov_fail(ctx, OVHIP_EINVAL, "ovhip_intra_level_launch: bad arguments", hipSuccess);
    LmcsWnd wnd;
    memset(&wnd, 0, sizeof(wnd));
    if (luts) { memcpy(wnd.bnd, luts->wnd_bnd, sizeof(wnd.bnd)); wnd.min_idx = luts->min_idx; wnd.max_idx = luts->max_idx; wnd.crs_offset = luts->crs_offset; }
    hipLaunchKernelGGL(k_intra_level, dim3(n, 1u << (geom & 3), (geom & 16) ? 2 : 1), dim3(64), 0, ctx->stream, *pic, *res, d_tasks, n, d_regions, wnd,
                       d_scales, log2_ctu_s);
    OV_LAUNCH_CHECK(ctx, "k_intra_level");
    return OVHIP_OK;
}

#ifdef OVHIP_CTU_PROBE
extern "C" int ovhip_debug_set_ctu_probe(void *d_buf)
{
    u64 *p = (u64 *)d_buf;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_probe), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#endif

// Words of the synchronisation block ovhip_intra_ctu_launch needs for a w x h picture (device memory, zeroed ONCE by the owner).
extern "C" size_t ovhip_intra_sync_words(int32_t width, int32_t height, int32_t log2_ctu_s)
{
    if (width <= 0 || height <= 0 || log2_ctu_s < 5 || log2_ctu_s > 7) return 0;
    const size_t s = (size_t)1 << log2_ctu_s;
    return SYNC_FLAGS + ((width + s - 1) / s) * ((height + s - 1) / s);
}

// The whole ordered pass in one launch (see k_intra_ctu).  d_tasks / d_ctus: DEVICE copies of ovhip_rec_itasks_by_ctu().
// d_sync: ovhip_intra_sync_words() words; epoch: a value this d_sync has not seen before and != 0 (a per-picture counter).
extern "C" int ovhip_intra_ctu_launch(ovhip_ctx *ctx, const ovhip_pic *pic, const ovhip_pic *res, const ovhip_itask *d_tasks, const ovhip_ictu *d_ctus,
                                      uint32_t n_ctus, const ovhip_lmcs_region *d_regions, const ovhip_lmcs_luts *luts, int16_t *d_scales,
                                      int32_t log2_ctu_s, uint32_t *d_sync, uint32_t epoch, uint32_t *abort_mirror)
{
    if (!ctx || !pic || !res) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    if (!n_ctus) return OVHIP_OK;
    if (!d_tasks || !d_ctus || !d_sync || !epoch || log2_ctu_s < 5 || log2_ctu_s > 7)
        return ov_fail(ctx, OVHIP_EINVAL, "ovhip_intra_ctu_launch: bad arguments", hipSuccess);
    if ((pic->w & 7) || (pic->h & 1) || (pic->stride_y & 3) || (pic->stride_c & 3) || ((uintptr_t)pic->y & 7) || ((uintptr_t)pic->cb & 7) || ((uintptr_t)pic->cr & 7))
        return ov_fail(ctx, OVHIP_EUNSUP, "ovhip_intra_ctu_launch: picture width must be a multiple of 8 and the planes 8-byte aligned", hipSuccess);
    LmcsWnd wnd;
    memset(&wnd, 0, sizeof(wnd));
    if (luts) { memcpy(wnd.bnd, luts->wnd_bnd, sizeof(wnd.bnd)); wnd.min_idx = luts->min_idx; wnd.max_idx = luts->max_idx; wnd.crs_offset = luts->crs_offset; }
    const int ncx = (pic->w + (1 << log2_ctu_s) - 1) >> log2_ctu_s;
    hipLaunchKernelGGL(k_intra_ctu, dim3(n_ctus), dim3(256), 0, ctx->stream, *pic, *res, d_tasks, d_ctus, d_regions, wnd, d_scales, log2_ctu_s, d_sync, epoch, ncx,
                       abort_mirror);
    OV_LAUNCH_CHECK(ctx, "k_intra_ctu");
    return OVHIP_OK;
}

// Items of the flow launch from the level-sorted tasks (HOST): one per (task, FSTRIP-sample strip, plane).  Returns the count
// (<= cap) or 0 when the picture cannot take this path (more tasks or strips than an item word holds).  (Until round 5 a prediction
// block less than a unit high -- horizontal ISP partitions of 1 or 2 rows, which share a state word with the partitions around them --
// sent the whole picture to the per-level launches; they now hand over inside the unit by tagged samples, see k_intra_flow.)
extern "C" size_t ovhip_intra_flow_items(const ovhip_itask *sorted, size_t n, uint32_t *items, size_t cap)
{
    size_t k = 0;
    for (size_t i = 0; sorted && i < n; ++i) {
        const ovhip_itask &t = sorted[i];
        if (i >= (1u << 24)) return 0;
        if (t.kind == OVHIP_IT_REGION) { if (k < cap) items[k] = (uint32_t)i; ++k; continue; }
        const int npx = 1 << (t.log2_w + t.log2_h), strips = (npx + FSTRIP - 1) / FSTRIP, comps = t.kind == OVHIP_IT_LUMA ? 1 : 2;
        if (strips > 32) return 0;                                   // the item word has five bits for the strip (a 64x64 block: 16)
        for (int st = 0; st < strips; ++st)
            for (int c = 0; c < comps; ++c) { if (k < cap) items[k] = (uint32_t)i | ((uint32_t)st << 24) | ((uint32_t)c << 29); ++k; }
    }
    return k <= cap ? k : 0;
}

// Words of the state block of ovhip_intra_flow_launch for a w x h picture (device memory, zeroed once by the owner).
extern "C" size_t ovhip_intra_flow_words(int32_t width, int32_t height)
{
    if (width <= 0 || height <= 0) return 0;
    return SYNC_FLAGS + 4 * (size_t)((width + 3) / 4) * ((height + 3) / 4);
}

// The whole ordered pass in one launch with per-unit dependency flags (k_intra_flow).  d_tasks: the level-sorted tasks;
// d_items: ovhip_intra_flow_items() of the same list (both DEVICE).  d_state: ovhip_intra_flow_words() words, zeroed once;
// epoch as for ovhip_intra_ctu_launch (d_state[0] = abort word, abort_mirror likewise).
extern "C" int ovhip_intra_flow_launch(ovhip_ctx *ctx, const ovhip_pic *pic, const ovhip_pic *res, const ovhip_itask *d_tasks, uint32_t n_tasks,
                                       const uint32_t *d_items, uint32_t n_items, const ovhip_lmcs_region *d_regions, const ovhip_lmcs_luts *luts,
                                       int16_t *d_scales, int32_t log2_ctu_s, uint32_t *d_state, uint32_t epoch, uint32_t *abort_mirror, int32_t prepare,
                                       int32_t n_workers)
{
    if (!ctx || !pic || !res) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    if (!n_items) return OVHIP_OK;
    if (!d_tasks || !d_items || !d_state || !epoch || epoch >= 0x7fffffffu || log2_ctu_s < 5 || log2_ctu_s > 7)
        return ov_fail(ctx, OVHIP_EINVAL, "ovhip_intra_flow_launch: bad arguments", hipSuccess);
    LmcsWnd wnd;
    memset(&wnd, 0, sizeof(wnd));
    if (luts) { memcpy(wnd.bnd, luts->wnd_bnd, sizeof(wnd.bnd)); wnd.min_idx = luts->min_idx; wnd.max_idx = luts->max_idx; wnd.crs_offset = luts->crs_offset; }
    const FlowState fs = flow_state_of(d_state, pic->w, pic->h);
    if (prepare) {
        hipLaunchKernelGGL(k_intra_flow_prepare, dim3((n_tasks + 15) / 16), dim3(256), 0, ctx->stream, d_tasks, n_tasks, fs, epoch);
        OV_LAUNCH_CHECK(ctx, "k_intra_flow_prepare");
    }
#ifdef OVHIP_CTU_PROBE
    if (hipMemcpyToSymbolAsync(HIP_SYMBOL(g_probe_items), &d_items, sizeof(d_items), 0, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return OVHIP_ELAUNCH;
#endif
    const int nap = 0;          // poll back-off variant: 0 = s_sleep 4 between polls (16 the same; 64 and exponential back-off measured 6 % slower)
    // n_workers workgroups take the items in turn (k_intra_flow); 0 or >= n_items: one workgroup per item
    const uint32_t grid = (n_workers > 0 && (uint32_t)n_workers < n_items) ? (uint32_t)n_workers : n_items;
    hipLaunchKernelGGL(k_intra_flow, dim3(grid), dim3(64), 0, ctx->stream, *pic, *res, d_tasks, d_items, n_items, d_regions, wnd, d_scales, log2_ctu_s, fs,
                       epoch, d_state, abort_mirror, nap);
    OV_LAUNCH_CHECK(ctx, "k_intra_flow");
    return OVHIP_OK;
}

// After the flow launches of a picture WITHOUT the inverse luma mapping (ovhip_lmcs_inverse_launch drops the hand-over bit of the
// luma samples as a side effect of its table lookup): clears it in the blocks of the luma tasks.  d_tasks: as ovhip_intra_flow_launch.
extern "C" int ovhip_intra_flow_untag_launch(ovhip_ctx *ctx, const ovhip_pic *pic, const ovhip_itask *d_tasks, uint32_t n_tasks, int32_t with_luma)
{
    if (!ctx || !pic) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    if (!n_tasks) return OVHIP_OK;
    if (!d_tasks) return ov_fail(ctx, OVHIP_EINVAL, "ovhip_intra_flow_untag_launch: bad arguments", hipSuccess);
    hipLaunchKernelGGL(k_flow_untag, dim3((n_tasks + 15) / 16), dim3(256), 0, ctx->stream, *pic, d_tasks, n_tasks, with_luma);
    OV_LAUNCH_CHECK(ctx, "k_flow_untag");
    return OVHIP_OK;
}
