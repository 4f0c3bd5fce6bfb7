// kernels_alf.hip -- K14 on gfx950: adaptive loop filter (4x4 Laplacian classification, 7x7 luma
// diamond, 5x5 chroma diamond, virtual-boundary handling) and cross-component ALF.
//
// Source = post-SAO picture, destination = output picture (ping-pong), so every tile is
// independent.  Luma: one 256-thread workgroup per 32x32 tile (a tile never straddles a CTU).  The
// 38x38 source window (3-sample halo, coordinates clamped at the picture border = the replicate
// padded filter_region of libovvc/rcn_ctu.c:361-508) is staged once in LDS; 64 lanes classify the
// tile's 64 4x4 blocks from LDS (8x8 Laplacian windows on the (r+c)-even lattice), then every lane
// filters 4 consecutive samples of one block row with that block's 13 coefficient/clip pairs.
// Chroma: 32x32 tile with a 2-sample halo in LDS, 5x5 diamond, then the CC-ALF term (7 luma taps
// read through L1/L2) added in registers before the single store.
//
// Replaces alf.classif, alf.luma[2], alf.chroma[2], alf.ccalf[2] and the driver
// rcn_alf_filter_line (libovvc/rcn_alf.c:283-1433).  Coefficient / clip sets are the arrays
// rcn_alf_reconstruct_coeff_APS() builds on the host (struct RCNALF, rcn_alf.h:60-66).
#include "ovvc_common.hip.h"

// Occupancy hint (waves per SIMD the register allocator must leave room for; 0 = compiler default).
// -DOV_WPE_ALF=n overrides it for sweeps.
#ifndef OV_WPE_ALF
#define OV_WPE_ALF 0
#endif
#if OV_WPE_ALF > 0
#define OV_OCC_ALF __attribute__((amdgpu_waves_per_eu(OV_WPE_ALF)))
#else
#define OV_OCC_ALF
#endif

namespace {

#define TL 32           /* tile size               */
#define LH 3            /* luma halo               */
#define LW (TL + 2 * LH)
#define LPAD 4          /* luma tile in LDS: 4 samples left of the tile (3 used) so that rows are 16-byte groups */
#define LWS 40          /* luma LDS row stride = 5 x 8 samples                                                   */
#define CH 2            /* chroma halo             */
#define CW (TL + 2 * CH)

// The rectangle a CTU's filter windows are clamped to: the picture, cut at the sides of the CTU that are borders of its rect entry
// (tile; ovhip_alf_ctu.border): the reference pads there exactly as at the picture border (rcn_extend_filter_region with the
// entry-local is_border, rcn_alf.c:1313-1318).  (x0, y0) = the CTU's origin, s = its size, in samples of the plane.
struct AlfRect { int x0, y0, x1, y1; };
__device__ __forceinline__ AlfRect alf_clamp_rect(int border, int x0, int y0, int s, int w, int h)
{
    AlfRect r = { 0, 0, w - 1, h - 1 };
    if (border & OVHIP_BORDER_LEFT) r.x0 = x0;
    if (border & OVHIP_BORDER_UPPER) r.y0 = y0;
    if (border & OVHIP_BORDER_RIGHT) r.x1 = min(r.x1, x0 + s - 1);
    if (border & OVHIP_BORDER_BOTTOM) r.y1 = min(r.y1, y0 + s - 1);
    return r;
}

__device__ __forceinline__ int alf_clipd(int clip, int ref, int a, int b)
{
    return ov_clip3(a - ref, -clip, clip) + ov_clip3(b - ref, -clip, clip);
}

// ---- packed form of one filter tap.  The two neighbours of a symmetric tap are picked out of the row-group registers
// straight into one dword (v_perm_b32); difference to the centre sample, both clips and the multiply-accumulate are
// then one packed instruction each (v_pk_sub_i16, v_pk_max_i16, v_pk_min_i16, v_dot2_i32_i16): 5 instructions per
// tap and sample instead of 8.  Differences of 10-bit samples and clip values up to 1 << 10 fit int16. ----
typedef short alf_s2 __attribute__((ext_vector_type(2)));
// group dword / half of column c: 6-dword groups start at column -4, 2-dword groups at column 0
#define A6I(c) (((c) + 4) >> 1)
#define A6H(c) (((c) + 4) & 1)
#define A2I(c) ((c) >> 1)
#define A2H(c) ((c) & 1)
// low half = half ha of dword da, high half = half hb of dword db
#define ALF_PK(da, ha, db, hb) __builtin_amdgcn_perm((db), (da), (uint32_t)((2 * (ha)) | ((2 * (ha) + 1) << 8) | ((4 + 2 * (hb)) << 16) | ((5 + 2 * (hb)) << 24)))
#define ALF_PK66(A, ca, B, cb) ALF_PK((A)[A6I(ca)], A6H(ca), (B)[A6I(cb)], A6H(cb))
#define ALF_PK22(A, ca, B, cb) ALF_PK((A)[A2I(ca)], A2H(ca), (B)[A2I(cb)], A2H(cb))
__device__ __forceinline__ uint32_t alf_dup(int v) { return __builtin_amdgcn_perm((uint32_t)v, (uint32_t)v, 0x01000100u); }   // (v, v) as int16 pair
__device__ __forceinline__ int alf_tap(uint32_t pair, uint32_t cur2, uint32_t c2, uint32_t nc2, uint32_t f2, int sum)
{
    alf_s2 d = __builtin_bit_cast(alf_s2, pair) - __builtin_bit_cast(alf_s2, cur2);
    d = __builtin_elementwise_min(__builtin_elementwise_max(d, __builtin_bit_cast(alf_s2, nc2)), __builtin_bit_cast(alf_s2, c2));
    return __builtin_amdgcn_sdot2(d, __builtin_bit_cast(alf_s2, f2), sum, false);
}
__device__ __forceinline__ int alf_tap_lin(uint32_t pair, uint32_t f2, int sum)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(alf_s2, pair), __builtin_bit_cast(alf_s2, f2), sum, false);
}

// alf_derive_filter_idx, rcn_alf.c:283-345
__device__ __forceinline__ void filter_idx(uint32_t sum_h, uint32_t sum_v, uint32_t sum_d, uint32_t sum_b, bool is_vb,
                                           int &cls, int &tr)
{
    const uint32_t scale = is_vb ? 96u : 64u;
    const int act = min((int)(((sum_h + sum_v) * scale) >> (OV_BD + 4)), 15);
    // th[] = {0,1,2,2,2,2,2,3,3,3,3,3,3,3,3,4}
    int c = act == 0 ? 0 : act == 1 ? 1 : act <= 6 ? 2 : act <= 14 ? 3 : 4;
    uint32_t max_hv, min_hv, max_db, min_db, max_dir, min_dir;
    int dir_hv, dir_db, main_dir, sec_dir;
    if (sum_v > sum_h) { max_hv = sum_v; min_hv = sum_h; dir_hv = 1; } else { max_hv = sum_h; min_hv = sum_v; dir_hv = 3; }
    if (sum_d > sum_b) { max_db = sum_d; min_db = sum_b; dir_db = 0; } else { max_db = sum_b; min_db = sum_d; dir_db = 2; }
    if (max_db * min_hv > max_hv * min_db) { max_dir = max_db; min_dir = min_db; main_dir = dir_db; sec_dir = dir_hv; }
    else { max_dir = max_hv; min_dir = min_hv; main_dir = dir_hv; sec_dir = dir_db; }
    if (max_dir * 2 > 9 * min_dir) c += (((main_dir & 1) << 1) + 2) * 5;
    else if (max_dir > 2 * min_dir) c += (((main_dir & 1) << 1) + 1) * 5;
    cls = c;
    const int k = (main_dir << 1) + (sec_dir >> 1);      // tr_lut = {0,1,0,2,2,3,1,3}
    tr = (0xDE84 >> (2 * k)) & 3;                        // packed 2-bit LUT
}

// one 32x32 luma tile (workgroup `tile0` of the luma part of k_alf)
__device__ __forceinline__ void alf_luma_tile(const ovhip_pic &dst, const ovhip_pic &src, const ovhip_alf_pic &alf, int nb_ctu_w,
                                              int tile0, uint16_t *s_t, uint8_t *s_cls, uint4 *s_sum, int row0, int row1)
{

    const int W = src.w, H = src.h;
    const int tid = threadIdx.x;
    const int ctu = 1 << alf.log2_ctu_s;
    const int ntx = (W + TL - 1) / TL, ntiles = ntx * ((H + TL - 1) / TL);
    for (int tile = tile0; tile < ntiles; tile = ntiles) {          // one pass; `continue` leaves the tile
    const int tx0 = (tile % ntx) * TL, ty0 = (tile / ntx) * TL;
    const ovhip_alf_ctu c = alf.ctus[(ty0 >> alf.log2_ctu_s) * nb_ctu_w + (tx0 >> alf.log2_ctu_s)];
    const bool on = c.flags & 4;

    // each lane owns 4 consecutive samples: block b = tid >> 2 (8x8 blocks of 4x4), row r = tid & 3
    const int b = tid >> 2, r = tid & 3;
    const int bx = (b & 7) * 4, by = (b >> 3) * 4;
    const int ox = tx0 + bx, oy = ty0 + by + r;

    if (!on) {          // ALF off for this CTU: plain copy (dst is a separate picture)
        if (oy < H && oy >= row0 && oy < row1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) if (ox + i < W) dst.y[oy * dst.stride_y + ox + i] = src.y[oy * src.stride_y + ox + i];
        }
        continue;
    }

    // 38 rows x 40 samples (tile + halo, 4 samples left so that the row is five 16-byte groups)
    const AlfRect rc = alf_clamp_rect(c.border, tx0 & ~(ctu - 1), ty0 & ~(ctu - 1), ctu, W, H);
    if (tx0 >= rc.x0 + LPAD && tx0 + TL + LPAD <= rc.x1 + 1 && ty0 >= rc.y0 + LH && ty0 + TL + LH <= rc.y1 + 1 && !(src.stride_y & 7)) {
        if (tid < LW * 5) {                                           // interior tile: one 16-byte load per lane
            const int yy = tid / 5, q = tid - yy * 5;
            *reinterpret_cast<uint4 *>(s_t + yy * LWS + 8 * q) =
                *reinterpret_cast<const uint4 *>(src.y + (ty0 + yy - LH) * src.stride_y + tx0 - LPAD + 8 * q);
        }
    } else {
        for (int i = tid; i < LW * LWS; i += 256) {                   // picture / rect-entry border: clamped coordinates
            const int yy = i / LWS, xx = i - yy * LWS;
            const int sy = ov_clip3(ty0 + yy - LH, rc.y0, rc.y1), sx = ov_clip3(tx0 + xx - LPAD, rc.x0, rc.x1);
            s_t[i] = src.y[sy * src.stride_y + sx];
        }
    }
    __syncthreads();

    // Virtual boundary as the reference derives it (rcn_alf.c:722, :1346, :1274-1283): CTU-local row
    // ctu - 4 for a full-height CTU, pic_h for a truncated one (then it only ever matches CTU-local rows
    // in pictures of a single CTU row); the VB filter variant is selected by check_virtual_bound().
    const int ctu_y0 = ty0 & ~(ctu - 1);
    const bool truncated = ctu_y0 + ctu > H;
    const int vbl = truncated ? H : ctu - 4;
    const int last_local = (ctu_y0 + (truncated ? H - ctu_y0 : ctu) - 1) & (ctu - 1);
    const bool req_vb = (last_local < vbl && last_local >= vbl - 4) || (last_local >= vbl && last_local <= vbl + 3);
    const int vb = ctu_y0 + vbl;               // same boundary in picture rows

    // ---- classification: 4 lanes per 4x4 block, one row pair of its 8x8 Laplacian window each ----
#define T(x, y) ((int)s_t[((y) + LH) * LWS + (x) + LPAD])
    {
        const int cb = tid >> 2, k = tid & 3;                    // block 0..63, row pair 0..3
        const int cbx = (cb & 7) * 4, cby = (cb >> 3) * 4;       // tile-local block origin
        const int rr = cby - 2 + 2 * k;                          // tile-local first row of the pair
        int above = rr - 1, below = rr + 2;
        if (ty0 + rr + 2 == vb) below = rr + 1;                  // pair (vb-2, vb-1): row vb is not available
        if (ty0 + rr == vb)     above = rr;                      // pair (vb, vb+1): row vb-1 is not available
        uint32_t sv = 0, sh = 0, sd = 0, sb = 0;
        {
            // the pair's 4 rows (above, rr, rr + 1, below), columns cbx-4 .. cbx+7, as 8-byte LDS reads
            uint32_t ra[6], r0[6], r1[6], rb[6];
            const uint16_t *base = s_t + LH * LWS + cbx;
            auto row3 = [&](int y, uint32_t d6[6]) {
                const uint2 *q2 = reinterpret_cast<const uint2 *>(base + y * LWS);
                const uint2 a = q2[0], b_ = q2[1], c_ = q2[2];
                d6[0] = a.x; d6[1] = a.y; d6[2] = b_.x; d6[3] = b_.y; d6[4] = c_.x; d6[5] = c_.y;
            };
            row3(above, ra); row3(rr, r0); row3(rr + 1, r1); row3(below, rb);
            // The two Laplacian positions of a step, (rr, c0) and (rr + 1, c0 + 1), ride in one dword: per direction the
            // neighbour sums are one v_pk_add_u16 and both |2 c - a - b| terms one v_sad_u16 that accumulates (all < 2^16).
            typedef unsigned short alf_u2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = 2 * q - 2, c1 = c0 + 1;             // columns relative to the block
                const alf_u2 ctr = __builtin_bit_cast(alf_u2, ALF_PK66(r0, c0, r1, c1));
                const uint32_t y2 = __builtin_bit_cast(uint32_t, ctr + ctr);
#define LAP(acc, A, ca, B, cb, C, cc_, D, cd)                                                                      \
                acc = __builtin_amdgcn_sad_u16(y2, __builtin_bit_cast(uint32_t, __builtin_bit_cast(alf_u2, ALF_PK66(A, ca, B, cb)) + \
                                                                               __builtin_bit_cast(alf_u2, ALF_PK66(C, cc_, D, cd))), acc)
                LAP(sv, ra, c0, r0, c1, r1, c0, rb, c1);
                LAP(sh, r0, c0 + 1, r1, c1 + 1, r0, c0 - 1, r1, c1 - 1);
                LAP(sd, ra, c0 - 1, r0, c1 - 1, r1, c0 + 1, rb, c1 + 1);
                LAP(sb, r1, c0 - 1, rb, c1 - 1, ra, c0 + 1, r0, c1 + 1);
#undef LAP
            }
        }
        // which pairs count: all four, or three next to the virtual boundary (rcn_alf.c:520-583)
        const int py = ty0 + cby;
        bool use = true, is_vb = false;
        if (py == vb - 4) { is_vb = true; use = k < 3; }
        if (py == vb)     { is_vb = true; use = k > 0; }
        if (!use) sv = sh = sd = sb = 0;
        // Every lane parks its row pair's sums; ONE wave then adds the four pairs of each of the tile's 64 blocks and derives
        // class and transpose (alf_derive_filter_idx is ~45 instructions whatever the number of active lanes: run by all
        // four waves with a quarter of their lanes it cost four times as much).
        (void)is_vb;
        s_sum[tid] = make_uint4(sv, sh, sd, sb);
    }
    __syncthreads();
    if (tid < 64) {
        const uint4 p0 = s_sum[4 * tid], p1 = s_sum[4 * tid + 1], p2 = s_sum[4 * tid + 2], p3 = s_sum[4 * tid + 3];
        const int py = ty0 + (tid >> 3) * 4;
        int cls, tr;
        filter_idx(p0.y + p1.y + p2.y + p3.y, p0.x + p1.x + p2.x + p3.x, p0.z + p1.z + p2.z + p3.z, p0.w + p1.w + p2.w + p3.w,
                   py == vb - 4 || py == vb, cls, tr);
        s_cls[tid] = (uint8_t)(cls | (tr << 5));
    }
    __syncthreads();

    if (oy >= H || oy < row0 || oy >= row1) continue;          // (row0, row1: the launch's row window; a 4x4 block lies inside or outside)
    const int ct = s_cls[b];
    const int cls = ct & 31, tr = ct >> 5;
    const int16_t *f = alf.luma_coeff + c.luma_set * OVHIP_ALF_LUMA_SET_SIZE + tr * 25 * 13 + cls * 13;
    const int16_t *cl = alf.luma_clip + c.luma_set * OVHIP_ALF_LUMA_SET_SIZE + tr * 25 * 13 + cls * 13;
    // the 12 coefficients / clip values of the class as 6 dwords each (rows of 13 int16 are only 2-byte aligned: the
    // loads are unaligned dword loads, which gfx9 global memory supports)
    typedef uint32_t alf_u32a2 __attribute__((aligned(2)));
    uint32_t fw[6], cw6[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) { fw[j] = *reinterpret_cast<const alf_u32a2 *>(f + 2 * j); cw6[j] = *reinterpret_cast<const alf_u32a2 *>(cl + 2 * j); }

    int d = 3;
    bool near = false;
    if (req_vb) {
        if (oy < vb && oy >= vb - 4) d = vb - 1 - oy;
        else if (oy >= vb && oy <= vb + 3) d = oy - vb;
        near = (oy == vb - 1) || (oy == vb);
    }
    const int o1 = min(d, 1), o2 = min(d, 2), o3 = min(d, 3);
    const int ly = by + r;
    // The lane's 4 outputs need columns bx-3 .. bx+6 of 7 rows: fetch the 12-sample groups bx-4 .. bx+7 of those
    // rows as 8-byte LDS reads (17 reads instead of ~100 two-byte ones) and pick samples out of registers.
    uint32_t r0[6], p1[6], m1[6], p2[6], m2[6], p3[2], m3[2];
    {
        const uint16_t *base = s_t + LH * LWS + bx;                   // column bx - 4 of tile row 0
        auto row3 = [&](int y, uint32_t d6[6]) {
            const uint2 *q = reinterpret_cast<const uint2 *>(base + y * LWS);
            const uint2 a = q[0], b_ = q[1], c_ = q[2];
            d6[0] = a.x; d6[1] = a.y; d6[2] = b_.x; d6[3] = b_.y; d6[4] = c_.x; d6[5] = c_.y;
        };
        auto row1 = [&](int y, uint32_t d2[2]) {
            const uint2 a = reinterpret_cast<const uint2 *>(base + y * LWS)[1];
            d2[0] = a.x; d2[1] = a.y;
        };
        row3(ly, r0); row3(ly + o1, p1); row3(ly - o1, m1); row3(ly + o2, p2); row3(ly - o2, m2);
        row1(ly + o3, p3); row1(ly - o3, m3);
    }
    // sample at column bx + c (c = -4 .. 7) of a 6-dword row group / (c = 0 .. 3) of a 2-dword group
#define S6(d6, c) ((((c) + 4) & 1) ? (int)((d6)[((c) + 4) >> 1] >> 16) : (int)((d6)[((c) + 4) >> 1] & 0xffff))
#define S2(d2, c) (((c) & 1) ? (int)((d2)[(c) >> 1] >> 16) : (int)((d2)[(c) >> 1] & 0xffff))
    // Clip value 1 << bitdepth never clips a 10-bit difference: filter sets without non-linear clipping (the 16
    // fixed sets, APS sets with alf_luma_clip_flag = 0, rcn_alf.c:196-240) take the linear form
    //   sum_i f_i * (a_i + b_i) - 2 * cur * sum_i f_i   (same integers, a third of the arithmetic)
    alf_s2 cm2 = __builtin_bit_cast(alf_s2, cw6[0]);
    int fsum = 0;
#pragma unroll
    for (int j = 1; j < 6; ++j) cm2 = __builtin_elementwise_min(cm2, __builtin_bit_cast(alf_s2, cw6[j]));
#pragma unroll
    for (int j = 0; j < 6; ++j) fsum = __builtin_amdgcn_sdot2(__builtin_bit_cast(alf_s2, fw[j]), (alf_s2)(1), fsum, false);
    const int cmin = min((int)cm2.x, (int)cm2.y);
    const bool linear = __all(cmin > OV_PIX_MAX);
    uint32_t f2[12], c2[12], nc2[12];                    // (f, f), (clip, clip), (-clip, -clip) as int16 pairs
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        f2[2 * j] = __builtin_amdgcn_perm(fw[j], fw[j], 0x01000100u); f2[2 * j + 1] = __builtin_amdgcn_perm(fw[j], fw[j], 0x03020302u);
    }
    if (!linear) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            c2[2 * j] = __builtin_amdgcn_perm(cw6[j], cw6[j], 0x01000100u); c2[2 * j + 1] = __builtin_amdgcn_perm(cw6[j], cw6[j], 0x03020302u);
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) nc2[i] = __builtin_bit_cast(uint32_t, (alf_s2)(0) - __builtin_bit_cast(alf_s2, c2[i]));
    }
    int outv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int cur = S6(r0, i);
        // the 12 symmetric neighbour pairs of sample i (same taps as alf_clipd's arguments above)
        const uint32_t n[12] = {
            ALF_PK22(p3, i, m3, i),
            ALF_PK66(p2, i + 1, m2, i - 1), ALF_PK66(p2, i, m2, i), ALF_PK66(p2, i - 1, m2, i + 1),
            ALF_PK66(p1, i + 2, m1, i - 2), ALF_PK66(p1, i + 1, m1, i - 1), ALF_PK66(p1, i, m1, i),
            ALF_PK66(p1, i - 1, m1, i + 1), ALF_PK66(p1, i - 2, m1, i + 2),
            ALF_PK66(r0, i + 3, r0, i - 3), ALF_PK66(r0, i + 2, r0, i - 2), ALF_PK66(r0, i + 1, r0, i - 1) };
        int sum;
        if (linear) {
            sum = -2 * cur * fsum;
#pragma unroll
            for (int k = 0; k < 12; ++k) sum = alf_tap_lin(n[k], f2[k], sum);
        } else {
            const uint32_t cur2 = ALF_PK66(r0, i, r0, i);
            sum = 0;
#pragma unroll
            for (int k = 0; k < 12; ++k) sum = alf_tap(n[k], cur2, c2[k], nc2[k], f2[k], sum);
        }
        sum = near ? (sum + 512) >> 10 : (sum + 64) >> 7;
        outv[i] = ov_clip_bd(sum + cur);
    }
#undef S6
#undef S2
    {
        uint16_t *o = dst.y + oy * dst.stride_y + ox;
        if (ox + 3 < W && !(dst.stride_y & 3)) {
            uint2 v; v.x = (uint32_t)outv[0] | ((uint32_t)outv[1] << 16); v.y = (uint32_t)outv[2] | ((uint32_t)outv[3] << 16);
            *reinterpret_cast<uint2 *>(o) = v;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) if (ox + i < W) o[i] = (uint16_t)outv[i];
        }
    }
    }
#undef T
}

// one 32x32 tile of chroma plane `comp` (1 Cb, 2 Cr)
__device__ __forceinline__ void alf_chroma_tile(const ovhip_pic &dst, const ovhip_pic &src, const ovhip_alf_pic &alf, int nb_ctu_w,
                                                int tile0, int comp, uint16_t *s_t, int row0, int row1)
{
    const int W = src.w, H = src.h, Wc = W >> 1, Hc = H >> 1;
    const int tid = threadIdx.x;
    const int ctu = 1 << alf.log2_ctu_s, ctuc = ctu >> 1;
    const int ntx = (Wc + TL - 1) / TL, ntiles = ntx * ((Hc + TL - 1) / TL);
    for (int tile = tile0; tile < ntiles; tile = ntiles) {
    const int tx0 = (tile % ntx) * TL, ty0 = (tile / ntx) * TL;
    const ovhip_alf_ctu c = alf.ctus[((ty0 * 2) >> alf.log2_ctu_s) * nb_ctu_w + ((tx0 * 2) >> alf.log2_ctu_s)];
    const bool on = c.flags & (comp == 1 ? 2 : 1);
    const int cc_idx = comp == 1 ? c.cc_cb_idx : c.cc_cr_idx;
    const uint16_t *sp = comp == 1 ? src.cb : src.cr;
    uint16_t *dp = comp == 1 ? dst.cb : dst.cr;

    const AlfRect rcc = alf_clamp_rect(c.border, tx0 & ~(ctuc - 1), ty0 & ~(ctuc - 1), ctuc, Wc, Hc);
    if (on) {
        // 36 rows x 40 samples (tile + halo, 4 samples left so that a row is five 16-byte groups)
        if (tx0 >= rcc.x0 + LPAD && tx0 + TL + LPAD <= rcc.x1 + 1 && ty0 >= rcc.y0 + CH && ty0 + TL + CH <= rcc.y1 + 1 && !(src.stride_c & 7)) {
            if (tid < CW * 5) {
                const int yy = tid / 5, q = tid - yy * 5;
                *reinterpret_cast<uint4 *>(s_t + yy * LWS + 8 * q) =
                    *reinterpret_cast<const uint4 *>(sp + (ty0 + yy - CH) * src.stride_c + tx0 - LPAD + 8 * q);
            }
        } else {
            for (int i = tid; i < CW * LWS; i += 256) {
                const int yy = i / LWS, xx = i - yy * LWS;
                const int sy = ov_clip3(ty0 + yy - CH, rcc.y0, rcc.y1), sx = ov_clip3(tx0 + xx - LPAD, rcc.x0, rcc.x1);
                s_t[i] = sp[sy * src.stride_c + sx];
            }
        }
        __syncthreads();
    }

    const int ctu_y0 = (ty0 * 2) & ~(ctu - 1);           // luma row of the CTU
    const bool truncated = ctu_y0 + ctu > H;
    int fc[6], cl[6];
    if (on) {
        const int alt = comp == 1 ? c.cb_alt : c.cr_alt;
#pragma unroll
        for (int i = 0; i < 6; ++i) { fc[i] = alf.chroma_coeff[alt * 7 + i]; cl[i] = alf.chroma_clip[alt * 7 + i]; }
    }
    int cf[7];
    if (cc_idx) {
#pragma unroll
        for (int i = 0; i < 7; ++i) cf[i] = alf.cc_coeff[((comp - 1) * 4 + (cc_idx - 1)) * 8 + i];
    }

    // lane: 4 consecutive samples of one row: row = tid >> 3 (0..31), x segment = (tid & 7) * 4
    const int ly = tid >> 3, lx0 = (tid & 7) * 4;
    const int oy = ty0 + ly;
    if (oy >= Hc || oy < (row0 >> 1) || oy >= (row1 >> 1)) continue;
    int o1 = 1, o2 = 2;
    bool near = false;
    if (on) {
        // alf_filter_cVB (always the VB variant: rcn_alf.c:1378-1386), vb compared with the CTU-local chroma row
        const int vb = truncated ? H / 2 : (ctu - 4) / 2;
        const int yl = oy & (ctuc - 1);
        int d = 2;
        if (yl < vb && yl >= vb - 2) d = vb - 1 - yl;
        else if (yl >= vb && yl <= vb + 1) d = yl - vb;
        near = (yl == vb - 1) || (yl == vb);
        o1 = min(d, 1); o2 = min(d, 2);
    }
    // CC-ALF vertical offsets (rcn_alf.c:758-772); vbPos is in LUMA rows for full CTUs, pic_h/2 for truncated ones
    int r1 = 1, r2 = -1, r3 = 2;
    if (cc_idx) {
        const int vbpos = truncated ? H / 2 : ctu - 4;
        const int pos = (oy << 1) & (ctu - 1);
        if (pos == vbpos - 2 || pos == vbpos + 1) r3 = r1;
        else if (pos == vbpos - 1 || pos == vbpos) r1 = r2 = r3 = 0;
    }
    const bool cc_inside = tx0 > rcc.x0 && ty0 > rcc.y0 && tx0 + TL <= rcc.x1 && ty0 + TL <= rcc.y1;
    const AlfRect rcl = { 2 * rcc.x0, 2 * rcc.y0, min(2 * rcc.x1 + 1, W - 1), min(2 * rcc.y1 + 1, H - 1) };      // the same rectangle in luma samples
    // 5 rows of the 12-sample group lx0-4 .. lx0+7 as 8-byte LDS reads (see alf_luma_tile)
    uint32_t r0[6], p1[6], m1[6], p2[2], m2[2];
    int cmin = 0x7fff, fsum = 0;
    if (on) {
        const uint16_t *base = s_t + CH * LWS + lx0;                 // column lx0 - 4 of tile row 0
        auto row3 = [&](int y, uint32_t d6[6]) {
            const uint2 *q = reinterpret_cast<const uint2 *>(base + y * LWS);
            const uint2 a = q[0], b_ = q[1], c_ = q[2];
            d6[0] = a.x; d6[1] = a.y; d6[2] = b_.x; d6[3] = b_.y; d6[4] = c_.x; d6[5] = c_.y;
        };
        auto row1 = [&](int y, uint32_t d2[2]) {
            const uint2 a = reinterpret_cast<const uint2 *>(base + y * LWS)[1];
            d2[0] = a.x; d2[1] = a.y;
        };
        row3(ly, r0); row3(ly + o1, p1); row3(ly - o1, m1); row1(ly + o2, p2); row1(ly - o2, m2);
#pragma unroll
        for (int i = 0; i < 6; ++i) { cmin = min(cmin, cl[i]); fsum += fc[i]; }
    }
    const bool linear = on && cmin > OV_PIX_MAX;                     // `on`, the filter and its clips are workgroup-uniform
    uint32_t f2[6], c2[6], nc2[6];                                   // (f, f), (clip, clip), (-clip, -clip) as int16 pairs
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        f2[i] = alf_dup(fc[i]);
        c2[i] = alf_dup(cl[i]);
        nc2[i] = __builtin_bit_cast(uint32_t, (alf_s2)(0) - __builtin_bit_cast(alf_s2, c2[i]));
    }
#define S6(d6, c) ((((c) + 4) & 1) ? (int)((d6)[((c) + 4) >> 1] >> 16) : (int)((d6)[((c) + 4) >> 1] & 0xffff))
#define S2(d2, c) (((c) & 1) ? (int)((d2)[(c) >> 1] >> 16) : (int)((d2)[(c) >> 1] & 0xffff))
    int outv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ox = min(tx0 + lx0 + i, Wc - 1);                    // columns past the picture are computed and dropped
        int out;
        if (on) {
            const int cur = S6(r0, i);
            int sum;
            const uint32_t n[6] = { ALF_PK22(p2, i, m2, i),
                                    ALF_PK66(p1, i + 1, m1, i - 1), ALF_PK66(p1, i, m1, i), ALF_PK66(p1, i - 1, m1, i + 1),
                                    ALF_PK66(r0, i + 2, r0, i - 2), ALF_PK66(r0, i + 1, r0, i - 1) };
            if (linear) {
                sum = -2 * cur * fsum;
#pragma unroll
                for (int k = 0; k < 6; ++k) sum = alf_tap_lin(n[k], f2[k], sum);
            } else {
                const uint32_t cur2 = ALF_PK66(r0, i, r0, i);
                sum = 0;
#pragma unroll
                for (int k = 0; k < 6; ++k) sum = alf_tap(n[k], cur2, c2[k], nc2[k], f2[k], sum);
            }
            sum = near ? (sum + 512) >> 10 : (sum + 64) >> 7;
            out = ov_clip_bd(sum + cur);
        } else {
            out = sp[oy * src.stride_c + ox];
        }
        if (cc_idx) {
            const int Lx = ox << 1, Ly = oy << 1;
            int cy, sum = 0;
            if (cc_inside) {                                            // tile away from the picture border: no clamping
                const uint16_t *lp = src.y + Ly * src.stride_y + Lx;
                cy = lp[0];
                sum += cf[0] * ((int)lp[r2 * src.stride_y] - cy);
                sum += cf[1] * ((int)lp[-1] - cy);
                sum += cf[2] * ((int)lp[1] - cy);
                sum += cf[3] * ((int)lp[r1 * src.stride_y - 1] - cy);
                sum += cf[4] * ((int)lp[r1 * src.stride_y] - cy);
                sum += cf[5] * ((int)lp[r1 * src.stride_y + 1] - cy);
                sum += cf[6] * ((int)lp[r3 * src.stride_y] - cy);
            } else {
#define LU(dx, dy) ((int)src.y[ov_clip3(Ly + (dy), rcl.y0, rcl.y1) * src.stride_y + ov_clip3(Lx + (dx), rcl.x0, rcl.x1)])
                cy = LU(0, 0);
                sum += cf[0] * (LU(0, r2) - cy);
                sum += cf[1] * (LU(-1, 0) - cy);
                sum += cf[2] * (LU(1, 0) - cy);
                sum += cf[3] * (LU(-1, r1) - cy);
                sum += cf[4] * (LU(0, r1) - cy);
                sum += cf[5] * (LU(1, r1) - cy);
                sum += cf[6] * (LU(0, r3) - cy);
#undef LU
            }
            sum = (sum + 64) >> 7;
            sum = ov_clip_bd(sum + (1 << OV_BD >> 1));
            out = ov_clip_bd(sum + out - (1 << OV_BD >> 1));
        }
        outv[i] = out;
    }
#undef S6
#undef S2
    uint16_t *o = dp + oy * dst.stride_c + tx0 + lx0;
    if (tx0 + lx0 + 3 < Wc && !(dst.stride_c & 3)) {
        uint2 v; v.x = (uint32_t)outv[0] | ((uint32_t)outv[1] << 16); v.y = (uint32_t)outv[2] | ((uint32_t)outv[3] << 16);
        *reinterpret_cast<uint2 *>(o) = v;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (tx0 + lx0 + i < Wc) o[i] = (uint16_t)outv[i];
    }
    }
}

// Luma and both chroma planes in ONE launch (they only share their input): workgroups [0, nl) take luma tiles,
// [nl, nl + 2 * nc) the Cb then Cr tiles.  One kernel boundary less, and the chroma tiles fill the luma tail.
__global__ __launch_bounds__(256) OV_OCC_ALF void k_alf(ovhip_pic dst, ovhip_pic src, ovhip_alf_pic alf, int nb_ctu_w, int nl, int nc, int t0_l, int t0_c, int row0, int row1)
{
    __shared__ __attribute__((aligned(16))) uint16_t s_t[(LW > CW ? LW : CW) * LWS];
    __shared__ uint8_t s_cls[64];
    __shared__ __attribute__((aligned(16))) uint4 s_sum[256];          // luma classification: one row pair's Laplacian sums per lane
    // XCD-aware tile order: every XCD takes one contiguous band of each plane's tiles (raster order), the same band of all three,
    // so the halo rows and columns neighbouring tiles share -- and the luma a chroma tile's CC-ALF taps read -- come out of ONE L2
    // (dealt round-robin, neighbouring tiles ran on different XCDs: every halo was fetched from memory once per XCD that needed it)
    const int b = blockIdx.x;
    // (t0_l / t0_c: first tile of the launch's row window in the luma / a chroma plane, ovhip_alf_launch_rows; 0 for a whole picture)
    if (b < nl) alf_luma_tile(dst, src, alf, nb_ctu_w, t0_l + (int)ov_xcd_slot_at(b, 0, nl), s_t, s_cls, s_sum, row0, row1);
    else {
        const int comp = 1 + (b - nl) / nc;
        alf_chroma_tile(dst, src, alf, nb_ctu_w, t0_c + (int)ov_xcd_slot_at(b, nl + (comp - 1) * nc, nc), comp, s_t, row0, row1);
    }
}

} // namespace

// Rows [row0, row1) of the picture (luma rows; both multiples of 8, or row1 = the picture's height): the band-wise picture job
// (ovvc_picture.hip).  Reads src (the SAO output) rows row0 - 3 .. row1 + 2 -- luma taps and the 4x4 classification windows; the
// chroma tiles' CC-ALF taps reach luma row row1 -- and writes dst rows [row0, row1) only: the tiles the window cuts are staged and
// classified whole (what lies outside it may not be final: its results are dropped), the stores of the rows outside are skipped.
extern "C" int ovhip_alf_launch_rows(ovhip_ctx *ctx, const ovhip_pic *dst, const ovhip_pic *src, const ovhip_alf_pic *alf, int32_t row0, int32_t row1)
{
    if (!ctx || !dst || !src || !alf) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    if (dst->w != src->w || dst->h != src->h || dst->y == src->y || alf->log2_ctu_s < 6 || alf->log2_ctu_s > 7 ||
        !alf->ctus || !alf->luma_coeff || !alf->luma_clip || !alf->chroma_coeff || !alf->chroma_clip || !alf->cc_coeff)
        return ov_fail(ctx, OVHIP_EINVAL, "ovhip_alf_launch: bad pictures / parameter tables", hipSuccess);
    if (row0 < 0 || row1 > src->h || row0 > row1 || (row0 & 7) || ((row1 & 7) && row1 != src->h))
        return ov_fail(ctx, OVHIP_EINVAL, "ovhip_alf_launch_rows: row window", hipSuccess);
    if (row0 == row1) return OVHIP_OK;
    const int nb_ctu_w = (src->w + (1 << alf->log2_ctu_s) - 1) >> alf->log2_ctu_s;
    const int ntx_l = (src->w + TL - 1) / TL, ntx_c = (src->w / 2 + TL - 1) / TL;
    const int ty0_l = row0 / TL, ty1_l = (row1 + TL - 1) / TL, ty0_c = (row0 / 2) / TL, ty1_c = (row1 / 2 + TL - 1) / TL;
    const int nl = ntx_l * (ty1_l - ty0_l), nc = ntx_c * (ty1_c - ty0_c);
    hipLaunchKernelGGL(k_alf, dim3(nl + 2 * nc), dim3(256), 0, ctx->stream, *dst, *src, *alf, nb_ctu_w, nl, nc, ty0_l * ntx_l, ty0_c * ntx_c, row0, row1);
    OV_LAUNCH_CHECK(ctx, "k_alf");
    return OVHIP_OK;
}

extern "C" int ovhip_alf_launch(ovhip_ctx *ctx, const ovhip_pic *dst, const ovhip_pic *src, const ovhip_alf_pic *alf)
{
    if (!src) return OVHIP_EINVAL;
    return ovhip_alf_launch_rows(ctx, dst, src, alf, 0, src->h);
}
