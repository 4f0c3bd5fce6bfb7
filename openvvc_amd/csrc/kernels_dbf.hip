// kernels_dbf.hip -- K12 on gfx950: VVC deblocking filter over picture-level edge planes.
//
// Two launches per picture: every vertical edge (luma + Cb + Cr), then every horizontal edge --
// the picture-level order the reference's per-CTU "vertical edges, then horizontal edges shifted
// 8 samples left" schedule reproduces (libovvc/rcn_df.c:2099-2106, :2169-2198).  One lane per
// 4-sample edge segment (2 chroma lines): the lane reads its 16-bit parameter word (bS, maximum
// filter lengths, average QP -- produced on the host by ovhip_rec_dbf_ctu), derives tc / beta
// (Table 43), evaluates the on/off + strong/weak + long-filter decisions on lines 0 and 3 and
// filters its 4 lines in place.  Segments of one direction never touch each other's samples
// (H.266 8.8.3: filter lengths are bounded by half the distance to the next edge), so all lanes
// run independently.  Horizontal edges map lanes to consecutive columns (coalesced rows);
// vertical edges read 16 samples per line per lane from the same rows as their neighbours (L1/L2).
//
// Replaces filter_vertical_edge / filter_horizontal_edge(_c), use_strong_filter_*, the 22 long
// filters filter_{h,v}_{3,5,7}_{3,5,7}, filter_luma_strong_small_*, filter_luma_weak_*,
// filter_chroma_{strong,weak}_* (libovvc/rcn_df.c:77-1148, :1433-1510, :2008-2085).
#include "ovvc_common.hip.h"
#define OVT_ATTR __device__
#include "vvc_dbf_tables.h"

// Occupancy hint (waves per SIMD the register allocator must leave room for; 0 = compiler default).
// -DOV_WPE_DBF=n overrides it for sweeps.
#ifndef OV_WPE_DBF
#define OV_WPE_DBF 0
#endif
#if OV_WPE_DBF > 0
#define OV_OCC_DBF __attribute__((amdgpu_waves_per_eu(OV_WPE_DBF)))
#else
#define OV_OCC_DBF
#endif

namespace {

struct Lim { int tc, beta; };

__device__ __forceinline__ Lim dbf_limits(int qp, int bs, int tc_off, int beta_off)
{
    Lim l;
    l.tc = ovt_dbf_tc[ov_clip3(qp + 2 * (bs - 1) + tc_off, 0, 66)];
    l.beta = (int)ovt_dbf_beta[ov_clip3(qp + beta_off, 0, 64)] << 2;
    return l;
}

// One line across the edge held in registers: s[8 + i] = q_i, s[7 - i] = p_i.
struct Line16 { int s[16]; };

template <int NP, int NQ>
__device__ __forceinline__ void load_line(const uint16_t *pix, int step, int *s)
{
#pragma unroll
    for (int i = 0; i < NP; ++i) s[7 - i] = pix[(-1 - i) * step];
#pragma unroll
    for (int i = 0; i < NQ; ++i) s[8 + i] = pix[i * step];
}
#define P(i) s[7 - (i)]
#define Q(i) s[8 + (i)]

__device__ __forceinline__ int dp_of(const int *s, int o) { return abs(P(2 + o) - 2 * P(1 + o) + P(0 + o)); }
__device__ __forceinline__ int dq_of(const int *s, int o) { return abs(Q(0 + o) - 2 * Q(1 + o) + Q(2 + o)); }

__device__ __forceinline__ bool strong_large(const int *s, int beta, int tc, int lp, int lq)
{
    int sp3 = abs(P(3) - P(0)), sq3 = abs(Q(3) - Q(0));
    if (lp == 7)      { sp3 += abs((P(4) - P(5)) - P(6) + P(7)); sp3 += abs(P(3) - P(7)) + 1; sp3 >>= 1; }
    else if (lp == 5) { sp3 += abs(P(3) - P(5)) + 1; sp3 >>= 1; }
    if (lq == 7)      { sq3 += abs((Q(4) - Q(5)) - Q(6) + Q(7)); sq3 += abs(Q(7) - Q(3)) + 1; sq3 >>= 1; }
    else if (lq == 5) { sq3 += abs(Q(5) - Q(3)) + 1; sq3 >>= 1; }
    return ((sp3 + sq3) < (beta * 3 >> 5)) && (abs(P(0) - Q(0)) < ((tc * 5 + 1) >> 1));
}

__device__ __forceinline__ bool strong_small(const int *s, int beta, int tc)
{
    return ((abs(P(3) - P(0)) + abs(Q(3) - Q(0))) < (beta >> 3)) && (abs(P(0) - Q(0)) < ((tc * 5 + 1) >> 1));
}

// the three line filters work on a line already in registers (s) and store only the samples they modify
__device__ __forceinline__ void long_regs(const int *s, uint16_t *pix, int step, int tc, int lp, int lq)
{
    const int f7[7] = { 59, 50, 41, 32, 23, 14, 5 }, f5[5] = { 58, 45, 32, 19, 6 }, f3[3] = { 53, 32, 11 };
    const int t7[7] = { 6, 5, 4, 3, 2, 1, 1 }, t3[3] = { 6, 4, 2 };
    const int ref_p = (P(lp - 1) + P(lp) + 1) >> 1, ref_q = (Q(lq - 1) + Q(lq) + 1) >> 1;
    int mid;
    if (lp == lq && lp == 7)
        mid = (2 * (P(0) + Q(0)) + P(1) + P(2) + P(3) + P(4) + P(5) + P(6) + Q(1) + Q(2) + Q(3) + Q(4) + Q(5) + Q(6) + 8) >> 4;
    else if (lp == lq)
        mid = (2 * (P(0) + P(1) + P(2) + Q(0) + Q(1) + Q(2)) + P(3) + P(4) + Q(3) + Q(4) + 8) >> 4;
    else if (lp + lq == 12)
        mid = (2 * (P(0) + P(1) + Q(0) + Q(1)) + P(2) + P(3) + P(4) + P(5) + Q(2) + Q(3) + Q(4) + Q(5) + 8) >> 4;
    else if (lp + lq == 8)
        mid = (P(0) + P(1) + P(2) + P(3) + Q(0) + Q(1) + Q(2) + Q(3) + 4) >> 3;
    else if (lp == 7)
        mid = (2 * (P(0) + Q(0)) + P(1) + P(2) + P(3) + P(4) + P(5) + P(6) + Q(0) + 3 * Q(1) + 2 * Q(2) + 8) >> 4;
    else
        mid = (2 * (P(0) + Q(0)) + Q(1) + Q(2) + Q(3) + Q(4) + Q(5) + Q(6) + P(0) + 3 * P(1) + 2 * P(2) + 8) >> 4;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        if (i < lp) {
            const int f = lp == 7 ? f7[i] : (lp == 5 ? f5[i < 5 ? i : 0] : f3[i < 3 ? i : 0]);
            const int cv = (tc * (lp == 3 ? t3[i < 3 ? i : 0] : t7[i])) >> 1;
            pix[(-1 - i) * step] = (uint16_t)ov_clip3((mid * f + ref_p * (64 - f) + 32) >> 6, P(i) - cv, P(i) + cv);
        }
        if (i < lq) {
            const int f = lq == 7 ? f7[i] : (lq == 5 ? f5[i < 5 ? i : 0] : f3[i < 3 ? i : 0]);
            const int cv = (tc * (lq == 3 ? t3[i < 3 ? i : 0] : t7[i])) >> 1;
            pix[i * step] = (uint16_t)ov_clip3((mid * f + ref_q * (64 - f) + 32) >> 6, Q(i) - cv, Q(i) + cv);
        }
    }
}

__device__ __forceinline__ void long_line(uint16_t *pix, int step, int tc, int lp, int lq)
{
    int s[16];
    load_line<8, 8>(pix, step, s);
    long_regs(s, pix, step, tc, lp, lq);
}

__device__ __forceinline__ void strong_regs(const int *s, uint16_t *pix, int step, int tc)
{
    const int p3 = P(3), p2 = P(2), p1 = P(1), p0 = P(0), q0 = Q(0), q1 = Q(1), q2 = Q(2), q3 = Q(3);
    pix[-3 * step] = (uint16_t)ov_clip3((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3, p2 - tc, p2 + tc);
    pix[-2 * step] = (uint16_t)ov_clip3((p2 + p1 + p0 + q0 + 2) >> 2, p1 - 2 * tc, p1 + 2 * tc);
    pix[-1 * step] = (uint16_t)ov_clip3((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3, p0 - 3 * tc, p0 + 3 * tc);
    pix[0]         = (uint16_t)ov_clip3((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3, q0 - 3 * tc, q0 + 3 * tc);
    pix[1 * step]  = (uint16_t)ov_clip3((p0 + q0 + q1 + q2 + 2) >> 2, q1 - 2 * tc, q1 + 2 * tc);
    pix[2 * step]  = (uint16_t)ov_clip3((p0 + q0 + q1 + 3 * q2 + 2 * q3 + 4) >> 3, q2 - tc, q2 + tc);
}

__device__ __forceinline__ void strong_line(uint16_t *pix, int step, int tc)
{
    int s[16];
    load_line<4, 4>(pix, step, s);
    strong_regs(s, pix, step, tc);
}

__device__ __forceinline__ void weak_regs(const int *s, uint16_t *pix, int step, int tc, bool ext_p, bool ext_q)
{
    const int p2 = P(2), p1 = P(1), p0 = P(0), q0 = Q(0), q1 = Q(1), q2 = Q(2);
    const int tc2p = ext_p ? tc >> 1 : 0, tc2q = ext_q ? tc >> 1 : 0;
    int delta = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
    if (abs(delta) < tc * 10) {
        delta = ov_clip3(delta, -tc, tc);
        const int d1 = ov_clip3((((p2 + p0 + 1) >> 1) - p1 + delta) >> 1, -tc2p, tc2p);
        const int d2 = ov_clip3((((q2 + q0 + 1) >> 1) - q1 - delta) >> 1, -tc2q, tc2q);
        pix[-2 * step] = (uint16_t)ov_clip_bd(p1 + d1);
        pix[-1 * step] = (uint16_t)ov_clip_bd(p0 + delta);
        pix[0]         = (uint16_t)ov_clip_bd(q0 - delta);
        pix[1 * step]  = (uint16_t)ov_clip_bd(q1 + d2);
    }
}

__device__ __forceinline__ void weak_line(uint16_t *pix, int step, int tc, bool ext_p, bool ext_q)
{
    int s[16];
    load_line<3, 3>(pix, step, s);
    weak_regs(s, pix, step, tc, ext_p, ext_q);
}

// ---- quad form (k_dbf_list): the 4 lanes of a quad hold the 4 lines of one segment, one line each, loaded ONCE;
// the decisions, which the reference takes on lines 0 and 3 (chroma: 0 and 1), travel by DPP quad broadcast.
// One memory round trip per segment instead of five (two decision lines, then four filtered lines one by one). ----
template <int SEL> __device__ __forceinline__ int quad_bcast(int v)
{
    return __builtin_amdgcn_mov_dpp(v, SEL * 0x55, 0xf, 0xf, true);      // quad_perm [SEL, SEL, SEL, SEL]
}

// line of NS samples each side of the edge; DIR 0 (vertical edge): the line is contiguous in memory and 8-byte
// aligned on both sides (x = 4 * ux), DIR 1: one sample per row -- a wave's lanes are adjacent columns
template <int DIR, int NS>
__device__ __forceinline__ void load_line_q(const uint16_t *pix, int step, int *s)
{
    if (DIR == 0 && !(reinterpret_cast<uintptr_t>(pix) & 7)) {
#pragma unroll
        for (int g = 0; g < NS / 4; ++g) {
            const uint2 a = *reinterpret_cast<const uint2 *>(pix - 4 * (g + 1));   // p(4g+3) .. p(4g)
            const uint2 b = *reinterpret_cast<const uint2 *>(pix + 4 * g);         // q(4g) .. q(4g+3)
            s[7 - (4 * g + 3)] = a.x & 0xffff; s[7 - (4 * g + 2)] = a.x >> 16; s[7 - (4 * g + 1)] = a.y & 0xffff; s[7 - 4 * g] = a.y >> 16;
            s[8 + 4 * g] = b.x & 0xffff; s[8 + 4 * g + 1] = b.x >> 16; s[8 + 4 * g + 2] = b.y & 0xffff; s[8 + 4 * g + 3] = b.y >> 16;
        }
    } else {
#pragma unroll
        for (int i = 0; i < NS; ++i) { s[7 - i] = pix[(-1 - i) * step]; s[8 + i] = pix[i * step]; }
    }
}

// filter_vertical_edge / filter_horizontal_edge (rcn_df.c:1433-1510, :2008-2085), lane = line `l` of the segment
template <int DIR>
__device__ __forceinline__ void luma_quad(uint16_t *pix, int step, Lim lim, int lp, int lq)
{
    const int beta = lim.beta, tc = lim.tc;
    const bool big = lp > 3 || lq > 3;
    int s[16];
    if (big) load_line_q<DIR, 8>(pix, step, s);
    else     load_line_q<DIR, 4>(pix, step, s);
    const int dp = dp_of(s, 0), dq = dq_of(s, 0);
    const int dp0 = quad_bcast<0>(dp), dq0 = quad_bcast<0>(dq), dp3 = quad_bcast<3>(dp), dq3 = quad_bcast<3>(dq);
    const int d0 = dp0 + dq0, d3 = dp3 + dq3;
    if (d0 + d3 >= beta) return;
    if (big) {
        int dpL = dp, dqL = dq;
        if (lp > 3) dpL = (dpL + dp_of(s, 3) + 1) >> 1;
        if (lq > 3) dqL = (dqL + dq_of(s, 3) + 1) >> 1;
        const int dL = dpL + dqL;
        const int ok = (dL < ((beta + 0x10) >> 5)) && strong_large(s, beta, tc, lp, lq);
        const int d0L = quad_bcast<0>(dL), d3L = quad_bcast<3>(dL);
        if ((d0L + d3L < beta) && quad_bcast<0>(ok) && quad_bcast<3>(ok)) { long_regs(s, pix, step, tc, lp, lq); return; }
    }
    const int oks = strong_small(s, beta, tc);
    const bool sw = lp > 2 && (d0 < ((beta + 4) >> 3)) && (d3 < ((beta + 4) >> 3)) && quad_bcast<0>(oks) && quad_bcast<3>(oks);
    if (sw) {
        strong_regs(s, pix, step, tc);
    } else {
        const int side = (beta + (beta >> 1)) >> 3;
        const bool ext_p = (dp0 + dp3) < side && lp > 1;
        const bool ext_q = (dq0 + dq3) < side && lp > 1;   // sic: max_l_p gates the Q side too (rcn_df.c:1505, :2080)
        weak_regs(s, pix, step, tc, ext_p, ext_q);
    }
}

// filter_veritcal_edge_c / filter_horizontal_edge_c (rcn_df.c:1107-1148, :1279-1319): 2 chroma lines per segment;
// lanes 2 and 3 of the quad shadow lanes 0 and 1 (same loads, no stores) so that the quad stays convergent
template <int DIR>
__device__ __forceinline__ void chroma_quad(uint16_t *pix, int step, Lim lim, bool large, bool ctb_b, bool store)
{
    const int tc = lim.tc, beta = lim.beta;
    if (tc == 0 || beta == 0) return;
    int s[16];
    load_line_q<DIR, 4>(pix, step, s);
    bool strong = false;
    if (large) {
        const int p3 = ctb_b ? P(1) : P(3);
        const int dpv = abs((ctb_b ? P(1) : P(2)) - 2 * P(1) + P(0));
        const int d = dpv + dq_of(s, 0);
        const int ok = ((abs(p3 - P(0)) + abs(Q(3) - Q(0))) < (beta >> 3)) && (abs(P(0) - Q(0)) < ((tc * 5 + 1) >> 1))
                       && (2 * d < (beta >> 2));
        const int da = quad_bcast<0>(d), db = quad_bcast<1>(d);
        strong = quad_bcast<0>(ok) && quad_bcast<1>(ok) && (da + db < beta);
    }
    if (!store) return;
    const int p3 = P(3), p2 = P(2), p1 = P(1), p0 = P(0), q0 = Q(0), q1 = Q(1), q2 = Q(2), q3 = Q(3);
    if (strong) {
        if (ctb_b) {
            pix[-1 * step] = (uint16_t)ov_clip3((3 * p1 + 2 * p0 + q0 + q1 + q2 + 4) >> 3, p0 - tc, p0 + tc);
            pix[0]         = (uint16_t)ov_clip3((2 * p1 + p0 + 2 * q0 + q1 + q2 + q3 + 4) >> 3, q0 - tc, q0 + tc);
        } else {
            pix[-3 * step] = (uint16_t)ov_clip3((3 * p3 + 2 * p2 + p1 + p0 + q0 + 4) >> 3, p2 - tc, p2 + tc);
            pix[-2 * step] = (uint16_t)ov_clip3((2 * p3 + p2 + 2 * p1 + p0 + q0 + q1 + 4) >> 3, p1 - tc, p1 + tc);
            pix[-1 * step] = (uint16_t)ov_clip3((p3 + p2 + p1 + 2 * p0 + q0 + q1 + q2 + 4) >> 3, p0 - tc, p0 + tc);
            pix[0]         = (uint16_t)ov_clip3((p2 + p1 + p0 + 2 * q0 + q1 + q2 + q3 + 4) >> 3, q0 - tc, q0 + tc);
        }
        pix[1 * step] = (uint16_t)ov_clip3((p1 + p0 + q0 + 2 * q1 + q2 + 2 * q3 + 4) >> 3, q1 - tc, q1 + tc);
        pix[2 * step] = (uint16_t)ov_clip3((p0 + q0 + q1 + 2 * q2 + 3 * q3 + 4) >> 3, q2 - tc, q2 + tc);
    } else {
        const int delta = ov_clip3(((q0 << 2) - (p0 << 2) + p1 - q1 + 4) >> 3, -tc, tc);
        pix[-1 * step] = (uint16_t)ov_clip_bd(p0 + delta);
        pix[0]         = (uint16_t)ov_clip_bd(q0 - delta);
    }
}

// filter_vertical_edge / filter_horizontal_edge (rcn_df.c:1433-1510, :2008-2085)
__device__ __forceinline__ void luma_segment(uint16_t *pix0, int step, int lstep, Lim lim, int lp, int lq)
{
    const int beta = lim.beta, tc = lim.tc;
    int a[16], b[16];                                 // lines 0 and 3
    if (lp > 3 || lq > 3) { load_line<8, 8>(pix0, step, a); load_line<8, 8>(pix0 + 3 * lstep, step, b); }
    else                  { load_line<4, 4>(pix0, step, a); load_line<4, 4>(pix0 + 3 * lstep, step, b); }
    const int dp0 = dp_of(a, 0), dq0 = dq_of(a, 0), dp3 = dp_of(b, 0), dq3 = dq_of(b, 0);
    const int d0 = dp0 + dq0, d3 = dp3 + dq3;
    if (d0 + d3 >= beta) return;
    bool sl = false;
    if (lp > 3 || lq > 3) {
        int dp0L = dp0, dq0L = dq0, dp3L = dp3, dq3L = dq3;
        if (lp > 3) { dp0L = (dp0L + dp_of(a, 3) + 1) >> 1; dp3L = (dp3L + dp_of(b, 3) + 1) >> 1; }
        if (lq > 3) { dq0L = (dq0L + dq_of(a, 3) + 1) >> 1; dq3L = (dq3L + dq_of(b, 3) + 1) >> 1; }
        const int d0L = dp0L + dq0L, d3L = dp3L + dq3L;
        sl = (d0L + d3L < beta) && (d0L < ((beta + 0x10) >> 5)) && (d3L < ((beta + 0x10) >> 5))
             && strong_large(a, beta, tc, lp, lq) && strong_large(b, beta, tc, lp, lq);
    }
    if (sl) {
#pragma unroll 1
        for (int l = 0; l < 4; ++l) long_line(pix0 + l * lstep, step, tc, lp, lq);
        return;
    }
    const bool sw = lp > 2 && (d0 < ((beta + 4) >> 3)) && (d3 < ((beta + 4) >> 3))
                    && strong_small(a, beta, tc) && strong_small(b, beta, tc);
    if (sw) {
#pragma unroll 1
        for (int l = 0; l < 4; ++l) strong_line(pix0 + l * lstep, step, tc);
    } else {
        const int side = (beta + (beta >> 1)) >> 3;
        const bool ext_p = (dp0 + dp3) < side && lp > 1;
        const bool ext_q = (dq0 + dq3) < side && lp > 1;   // sic: max_l_p gates the Q side too (rcn_df.c:1505, :2080)
#pragma unroll 1
        for (int l = 0; l < 4; ++l) weak_line(pix0 + l * lstep, step, tc, ext_p, ext_q);
    }
}

// filter_veritcal_edge_c / filter_horizontal_edge_c (rcn_df.c:1107-1148, :1279-1319): 2 chroma lines
__device__ __forceinline__ void chroma_segment(uint16_t *pix0, int step, int lstep, Lim lim, bool large, bool ctb_b)
{
    const int tc = lim.tc, beta = lim.beta;
    if (tc == 0 || beta == 0) return;
    int s0[16], s1[16];
    load_line<4, 4>(pix0, step, s0);
    load_line<4, 4>(pix0 + lstep, step, s1);
    bool strong = false;
    if (large) {
        int d[2];
        bool ok = true;
#pragma unroll
        for (int l = 0; l < 2; ++l) {
            const int *s = l ? s1 : s0;
            const int p3 = ctb_b ? P(1) : P(3);
            const int dp = abs((ctb_b ? P(1) : P(2)) - 2 * P(1) + P(0));
            d[l] = dp + dq_of(s, 0);
            ok = ok && ((abs(p3 - P(0)) + abs(Q(3) - Q(0))) < (beta >> 3)) && (abs(P(0) - Q(0)) < ((tc * 5 + 1) >> 1));
        }
        strong = ok && (d[0] + d[1] < beta) && (2 * d[0] < (beta >> 2)) && (2 * d[1] < (beta >> 2));
    }
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        const int *s = l ? s1 : s0;
        uint16_t *pix = pix0 + l * lstep;
        const int p3 = P(3), p2 = P(2), p1 = P(1), p0 = P(0), q0 = Q(0), q1 = Q(1), q2 = Q(2), q3 = Q(3);
        if (strong) {
            if (ctb_b) {
                pix[-1 * step] = (uint16_t)ov_clip3((3 * p1 + 2 * p0 + q0 + q1 + q2 + 4) >> 3, p0 - tc, p0 + tc);
                pix[0]         = (uint16_t)ov_clip3((2 * p1 + p0 + 2 * q0 + q1 + q2 + q3 + 4) >> 3, q0 - tc, q0 + tc);
            } else {
                pix[-3 * step] = (uint16_t)ov_clip3((3 * p3 + 2 * p2 + p1 + p0 + q0 + 4) >> 3, p2 - tc, p2 + tc);
                pix[-2 * step] = (uint16_t)ov_clip3((2 * p3 + p2 + 2 * p1 + p0 + q0 + q1 + 4) >> 3, p1 - tc, p1 + tc);
                pix[-1 * step] = (uint16_t)ov_clip3((p3 + p2 + p1 + 2 * p0 + q0 + q1 + q2 + 4) >> 3, p0 - tc, p0 + tc);
                pix[0]         = (uint16_t)ov_clip3((p2 + p1 + p0 + 2 * q0 + q1 + q2 + q3 + 4) >> 3, q0 - tc, q0 + tc);
            }
            pix[1 * step] = (uint16_t)ov_clip3((p1 + p0 + q0 + 2 * q1 + q2 + 2 * q3 + 4) >> 3, q1 - tc, q1 + tc);
            pix[2 * step] = (uint16_t)ov_clip3((p0 + q0 + q1 + 2 * q2 + 3 * q3 + 4) >> 3, q2 - tc, q2 + tc);
        } else {
            const int delta = ov_clip3(((q0 << 2) - (p0 << 2) + p1 - q1 + 4) >> 3, -tc, tc);
            pix[-1 * step] = (uint16_t)ov_clip_bd(p0 + delta);
            pix[0]         = (uint16_t)ov_clip_bd(q0 - delta);
        }
    }
}
#undef P
#undef Q

// DIR 0: vertical edges, DIR 1: horizontal edges.  blockIdx.y: 0 luma, 1 Cb, 2 Cr.
template <int DIR>
__global__ __launch_bounds__(256) void k_dbf(ovhip_pic pic, ovhip_dbf_planes pl)
{
    const int comp = blockIdx.y;
    const int w4 = pl.w4, h4 = pl.h4;
    const int nthreads = gridDim.x * 256;
    // every lane strides over the edge-parameter plane (most words are 0 = no edge; k_dbf_list takes the compact lists)
    if (comp == 0) {
        for (int tid = blockIdx.x * 256 + threadIdx.x; tid < w4 * h4; tid += nthreads) {
            const int ux = tid % w4, uy = tid / w4;
            const int v = (DIR ? pl.luma_h : pl.luma_v)[tid];
            if (!(v & 3)) continue;
            // never filter across the picture boundary (the recorder does not emit such edges)
            if ((DIR ? uy : ux) == 0) continue;
            const Lim lim = dbf_limits(v >> 8, v & 3, pl.tc_offset, pl.beta_offset);
            if (!(lim.tc || lim.beta)) continue;
            uint16_t *p = pic.y + (uy * 4) * pic.stride_y + ux * 4;
            luma_segment(p, DIR ? pic.stride_y : 1, DIR ? 1 : pic.stride_y, lim, (v >> 2) & 7, (v >> 5) & 7);
        }
    } else {
        // chroma edge planes: vertical [h4][w4c] (every second unit column), horizontal [h4c][w4]
        const int cw = DIR ? w4 : (w4 + 1) >> 1, chh = DIR ? (h4 + 1) >> 1 : h4;
        const uint16_t *plane = DIR ? (comp == 1 ? pl.cb_h : pl.cr_h) : (comp == 1 ? pl.cb_v : pl.cr_v);
        for (int tid = blockIdx.x * 256 + threadIdx.x; tid < cw * chh; tid += nthreads) {
            const int cx = tid % cw, cy = tid / cw;
            const int v = plane[tid];
            if (!(v & OVHIP_DBF_C_ON)) continue;
            const int ux = DIR ? cx : cx * 2, uy = DIR ? cy * 2 : cy;
            if ((DIR ? uy : ux) == 0) continue;
            const Lim lim = dbf_limits(v >> 8, 1 + !!(v & OVHIP_DBF_C_BS2), pl.tc_offset, pl.beta_offset);
            uint16_t *p = (comp == 1 ? pic.cb : pic.cr) + (uy * 2) * pic.stride_c + ux * 2;
            chroma_segment(p, DIR ? pic.stride_c : 1, DIR ? 1 : pic.stride_c, lim, v & OVHIP_DBF_C_LARGE, v & OVHIP_DBF_C_CTB_B);
        }
    }
}

// The same filter over the compact edge lists (ovhip_dbf_compact): 4 lanes per edge, one per line of the segment.
template <int DIR>
__global__ __launch_bounds__(256) OV_OCC_DBF void k_dbf_list(ovhip_pic pic, const ovhip_dbf_edge *__restrict__ edges, uint32_t n,
                                                  uint64_t tc_pack, uint64_t beta_pack)
{
    // XCD-aware order (ov_xcd_slot): the list is in decoding order, an XCD takes one contiguous chunk of it = a band of the picture
    const uint32_t tid = ov_xcd_slot(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    const uint32_t ei = tid >> 2;
    const int l = tid & 3;
    if (ei >= n) return;                                   // whole quads leave together
    const ovhip_dbf_edge e = edges[ei];
    const int v = e.word;
    // slice-level offsets (DBFInfo.tc_offset / beta_offset): 8 signed bytes each, the edge carries the index
    const int osh = (e.pad & 7) * 8;
    const int tc_offset = (int8_t)(tc_pack >> osh), beta_offset = (int8_t)(beta_pack >> osh);
    if (e.comp == 0) {
        const Lim lim = dbf_limits(v >> 8, v & 3, tc_offset, beta_offset);
        if (!(lim.tc || lim.beta)) return;
        uint16_t *p = pic.y + (e.uy * 4) * pic.stride_y + e.ux * 4 + l * (DIR ? 1 : pic.stride_y);
        luma_quad<DIR>(p, DIR ? pic.stride_y : 1, lim, (v >> 2) & 7, (v >> 5) & 7);
    } else {
        const Lim lim = dbf_limits(v >> 8, 1 + !!(v & OVHIP_DBF_C_BS2), tc_offset, beta_offset);
        uint16_t *p = (e.comp == 1 ? pic.cb : pic.cr) + (e.uy * 2) * pic.stride_c + e.ux * 2 + (l & 1) * (DIR ? 1 : pic.stride_c);
        chroma_quad<DIR>(p, DIR ? pic.stride_c : 1, lim, v & OVHIP_DBF_C_LARGE, v & OVHIP_DBF_C_CTB_B, l < 2);
    }
}

} // namespace

extern "C" int ovhip_dbf_launch_edges_ex(ovhip_ctx *ctx, const ovhip_pic *pic, const ovhip_dbf_edge *d_edges_v, uint32_t n_v,
                                         const ovhip_dbf_edge *d_edges_h, uint32_t n_h, const ovhip_dbf_offsets *offsets)
{
    if (!ctx || !pic || !offsets) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    if ((n_v && !d_edges_v) || (n_h && !d_edges_h))
        return ov_fail(ctx, OVHIP_EINVAL, "ovhip_dbf_launch_edges: null edge list", hipSuccess);
    uint64_t tc_pack, beta_pack;
    memcpy(&tc_pack, offsets->tc, 8); memcpy(&beta_pack, offsets->beta, 8);
    if (n_v) {
        hipLaunchKernelGGL(k_dbf_list<0>, dim3((n_v + 63) / 64), dim3(256), 0, ctx->stream, *pic, d_edges_v, n_v, tc_pack, beta_pack);
        OV_LAUNCH_CHECK(ctx, "k_dbf_list<v>");
    }
    if (n_h) {
        hipLaunchKernelGGL(k_dbf_list<1>, dim3((n_h + 63) / 64), dim3(256), 0, ctx->stream, *pic, d_edges_h, n_h, tc_pack, beta_pack);
        OV_LAUNCH_CHECK(ctx, "k_dbf_list<h>");
    }
    return OVHIP_OK;
}

extern "C" int ovhip_dbf_launch_edges(ovhip_ctx *ctx, const ovhip_pic *pic, const ovhip_dbf_edge *d_edges_v, uint32_t n_v,
                                      const ovhip_dbf_edge *d_edges_h, uint32_t n_h, int32_t beta_offset, int32_t tc_offset)
{
    if (beta_offset < -128 || beta_offset > 127 || tc_offset < -128 || tc_offset > 127) return OVHIP_EINVAL;
    ovhip_dbf_offsets o;
    for (int i = 0; i < OVHIP_DBF_MAX_OFFSETS; ++i) { o.beta[i] = (int8_t)beta_offset; o.tc[i] = (int8_t)tc_offset; }
    return ovhip_dbf_launch_edges_ex(ctx, pic, d_edges_v, n_v, d_edges_h, n_h, &o);
}

extern "C" int ovhip_dbf_launch(ovhip_ctx *ctx, const ovhip_pic *pic, const ovhip_dbf_planes *pl)
{
    if (!ctx || !pic || !pl) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    if (!pl->luma_v || !pl->luma_h || !pl->cb_v || !pl->cr_v || !pl->cb_h || !pl->cr_h)
        return ov_fail(ctx, OVHIP_EINVAL, "ovhip_dbf_launch: null edge plane", hipSuccess);
    if (pl->w4 != (pic->w + 3) / 4 || pl->h4 != (pic->h + 3) / 4)
        return ov_fail(ctx, OVHIP_EINVAL, "ovhip_dbf_launch: edge planes do not match the picture", hipSuccess);
    const int n = pl->w4 * pl->h4;
    const int nb = (n + 255) / 256;
    dim3 grid(nb, 3);                            // one lane per edge word measured faster than a capped resident grid
    hipLaunchKernelGGL(k_dbf<0>, grid, dim3(256), 0, ctx->stream, *pic, *pl);
    OV_LAUNCH_CHECK(ctx, "k_dbf<v>");
    hipLaunchKernelGGL(k_dbf<1>, grid, dim3(256), 0, ctx->stream, *pic, *pl);
    OV_LAUNCH_CHECK(ctx, "k_dbf<h>");
    return OVHIP_OK;
}
