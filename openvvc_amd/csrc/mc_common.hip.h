// mc_common.hip.h -- device building blocks shared by the plain (kernels_mc.hip) and the refined
// BDOF / DMVR (kernels_mcx.hip) motion-compensation kernels: reference-window staging, packed
// v_dot2 FIR, separable H (LDS->LDS, transposed) and V (LDS->registers) passes.
#pragma once
#include "ovvc_common.hip.h"
#define OVT_ATTR __device__
#include "vvc_mc_taps.h"

namespace {

#define MC_MAX_REFS 16
struct RefTable { ovhip_pic p[MC_MAX_REFS]; };

typedef short short2v __attribute__((ext_vector_type(2)));

#define WIN_STRIDE 28   /* luma window row in LDS: 7 aligned qwords (<= 3 + 23 samples) = 56 B            */
#define CWIN_STRIDE 16  /* chroma window row: 4 aligned qwords (<= 3 + 11 samples) = 32 B                 */
#define LUMA_WIN (23 * WIN_STRIDE)
#define CHR_WIN  (11 * CWIN_STRIDE)
#define HT_STRIDE 28    /* transposed H-pass tile: one COLUMN per row of HT_STRIDE int16 (h + 7 <= 23), 8-B aligned rows */
#define CHT_STRIDE 12

// ---- stage 1: reference window -> LDS.  Every lane loads one ALIGNED 8-byte group of 4 samples -- QW lanes per
// window row, 64/QW rows per instruction: 3 instructions for a 23x23 luma window, 1 for an 11x11 chroma window --
// and parks it with one ds_write_b64; the sub-group offset `off` (0..3 samples) is resolved by the horizontal pass.
// Three ways to fill the registers, all equal to emulate_block_border() (rcn_inter.c:148-225) where it applies:
//   interior  window inside the picture: plain loads;
//   clamped   window crosses a border of a picture whose width and stride are multiples of 4 samples (every VVC
//             4:2:0 picture in buffers of this library): rows clamp per lane, and an aligned group lies either
//             wholly inside or wholly outside the picture, so an outside group is the nearest inside group's edge
//             sample replicated -- same loads as the interior path, no extra latency;
//   slow      any other geometry: per-sample loads with clamped coordinates, parked at off = 0. ----
template <int QW, int NIT, int COLS, int NITS>
struct WinStage {
    uint2 q[NIT];
    bool fast;
    int off, side;          // side: -1 / +1 = this lane's groups lie left / right of the picture

    // (bitwise on purpose: short-circuit && on wave-uniform values turns into scalar branches, and scalar issue is what
    // bounds k_mc2)
    static __device__ __forceinline__ int interior(int sx0, int sy0, int ww, int wh, int rw, int rh)
    {
        const int ax = sx0 & ~3, nq = (sx0 - ax + ww + 3) >> 2;
        return (int)(ax >= 0) & (int)(ax + 4 * nq <= rw) & (int)(sy0 >= 0) & (int)(sy0 + wh <= rh);
    }
    __device__ __forceinline__ void issue_fast(const uint16_t *__restrict__ ref, int rstride, int sx0, int sy0, int ww, int wh, int lane)
    {
        const int ax = sx0 & ~3;
        off = sx0 - ax; fast = true; side = 0;
        const int nq = (off + ww + 3) >> 2;
        const int c = lane & (QW - 1), r0 = lane / QW;
        const uint16_t *base = ref + ov_rowoff(sy0 + r0, rstride) + ax + 4 * c;
#pragma unroll
        for (int k = 0; k < NIT; ++k)
            if (c < nq && (64 / QW) * k + r0 < wh) q[k] = *reinterpret_cast<const uint2 *>(base + (64 / QW) * k * rstride);
    }
    __device__ __forceinline__ void issue_clamped(const uint16_t *__restrict__ ref, int rstride, int rw, int rh, int sx0, int sy0,
                                                  int ww, int wh, int lane)
    {
        const int ax = sx0 & ~3;
        off = sx0 - ax; fast = true;
        const int nq = (off + ww + 3) >> 2;
        const int c = lane & (QW - 1), r0 = lane / QW;
        const int qx = ax + 4 * c;
        side = qx < 0 ? -1 : qx >= rw ? 1 : 0;
        const uint16_t *base = ref + ov_clip3(qx, 0, rw - 4);
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int r = (64 / QW) * k + r0;
            if (c < nq && r < wh) q[k] = *reinterpret_cast<const uint2 *>(base + ov_rowoff(ov_clip3(sy0 + r, 0, rh - 1), rstride));
        }
    }
    __device__ __forceinline__ void issue_slow(const uint16_t *__restrict__ ref, int rstride, int rw, int rh, int sx0, int sy0,
                                               int ww, int wh, int lane, uint16_t *s_win, int wstride)
    {
        off = 0; fast = false; side = 0;
        const int c = lane & (COLS - 1), r0 = lane / COLS;
        const int sx = ov_clip3(sx0 + c, 0, rw - 1);
#pragma unroll 1
        for (int k = 0; k < NITS; ++k) {
            const int r = (64 / COLS) * k + r0;
            if (c < ww && r < wh) s_win[r * wstride + c] = ref[ov_clip3(sy0 + r, 0, rh - 1) * rstride + sx];
        }
    }
    template <bool FIX = true>
    __device__ __forceinline__ void park(uint16_t *s_win, int wstride, int ww, int wh, int lane) const
    {
        if (fast) {
            const int nq = (off + ww + 3) >> 2;
            const int c = lane & (QW - 1), r0 = lane / QW;
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                const int r = (64 / QW) * k + r0;
                if (c < nq && r < wh) {
                    uint2 v = q[k];
                    if (FIX && side) { const uint32_t e = (side < 0 ? v.x & 0xffffu : v.y >> 16) * 0x10001u; v.x = e; v.y = e; }
                    *reinterpret_cast<uint2 *>(s_win + r * wstride + 4 * c) = v;
                }
            }
        }
    }
};
typedef WinStage<8, 3, 32, 12> LumaStage;
typedef WinStage<4, 1, 16, 3> ChromaStage;

// All reference windows of one unit (<= 16x16 luma, both lists, Cb and Cr) -> LDS.  The path is chosen once per unit:
// every window interior -> all loads issued, then all parked; a border unit goes list by list (rolled, so that the
// rare path does not raise the kernel's register budget).  `geom` is the geometry shared by dst and every reference.
// lwin[l] / cwin[plane * 2 + l] are the LDS windows; offl / offc receive the sub-group offsets.
__device__ __forceinline__ void stage_unit_windows(const ovhip_pic &geom, const uint16_t *const ry[2], const uint16_t *const rcb[2],
                                                   const uint16_t *const rcr[2], const int lx[2], const int ly[2],
                                                   const int cx[2], const int cy[2], int w, int h, int dir, bool do_l, bool do_c,
                                                   int lane, uint16_t *const lwin[2], int lstride, uint16_t *const cwin[4], int cstride,
                                                   int offl[2], int offc[2])
{
    const int wc = w >> 1, hc = h >> 1, pw = geom.w, ph = geom.h, pwc = geom.w >> 1, phc = geom.h >> 1;
    const bool aligned = !((geom.stride_y | geom.stride_c | pw | pwc) & 3);
    int all_in = aligned;
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        const int need_l = ((dir >> l) & 1) & (int)do_l, need_c = ((dir >> l) & 1) & (int)do_c;
        all_in &= (need_l ^ 1) | LumaStage::interior(lx[l], ly[l], w + 7, h + 7, pw, ph);
        all_in &= (need_c ^ 1) | ChromaStage::interior(cx[l], cy[l], wc + 3, hc + 3, pwc, phc);
    }
    const bool fast = all_in;
    offl[0] = offl[1] = offc[0] = offc[1] = 0;
    if (fast) {
        LumaStage sl[2];
        ChromaStage sc[2][2];
#pragma unroll
        for (int l = 0; l < 2; ++l) {
            if (!(dir & (1 << l))) continue;
            if (do_l) sl[l].issue_fast(ry[l], geom.stride_y, lx[l], ly[l], w + 7, h + 7, lane);
            if (do_c) {
                sc[0][l].issue_fast(rcb[l], geom.stride_c, cx[l], cy[l], wc + 3, hc + 3, lane);
                sc[1][l].issue_fast(rcr[l], geom.stride_c, cx[l], cy[l], wc + 3, hc + 3, lane);
            }
        }
#pragma unroll
        for (int l = 0; l < 2; ++l) {
            if (!(dir & (1 << l))) continue;
            if (do_l) { sl[l].park<false>(lwin[l], lstride, w + 7, h + 7, lane); offl[l] = sl[l].off; }
            if (do_c) {
                sc[0][l].park<false>(cwin[l], cstride, wc + 3, hc + 3, lane);
                sc[1][l].park<false>(cwin[2 + l], cstride, wc + 3, hc + 3, lane);
                offc[l] = sc[0][l].off;
            }
        }
        return;
    }
#pragma unroll 1
    for (int l = 0; l < 2; ++l) {
        if (!(dir & (1 << l))) continue;
        const uint16_t *y = l ? ry[1] : ry[0], *cb = l ? rcb[1] : rcb[0], *cr = l ? rcr[1] : rcr[0];
        const int x0 = l ? lx[1] : lx[0], y0 = l ? ly[1] : ly[0], xc = l ? cx[1] : cx[0], yc = l ? cy[1] : cy[0];
        uint16_t *wl = l ? lwin[1] : lwin[0], *wb = l ? cwin[1] : cwin[0], *wr = l ? cwin[3] : cwin[2];
        LumaStage a;
        ChromaStage b0, b1;
        a.off = b0.off = 0;
        if (aligned) {
            if (do_l) a.issue_clamped(y, geom.stride_y, pw, ph, x0, y0, w + 7, h + 7, lane);
            if (do_c) {
                b0.issue_clamped(cb, geom.stride_c, pwc, phc, xc, yc, wc + 3, hc + 3, lane);
                b1.issue_clamped(cr, geom.stride_c, pwc, phc, xc, yc, wc + 3, hc + 3, lane);
            }
            if (do_l) a.park<true>(wl, lstride, w + 7, h + 7, lane);
            if (do_c) { b0.park<true>(wb, cstride, wc + 3, hc + 3, lane); b1.park<true>(wr, cstride, wc + 3, hc + 3, lane); }
        } else {
            if (do_l) a.issue_slow(y, geom.stride_y, pw, ph, x0, y0, w + 7, h + 7, lane, wl, lstride);
            if (do_c) {
                b0.issue_slow(cb, geom.stride_c, pwc, phc, xc, yc, wc + 3, hc + 3, lane, wb, cstride);
                b1.issue_slow(cr, geom.stride_c, pwc, phc, xc, yc, wc + 3, hc + 3, lane, wr, cstride);
            }
        }
        if (l) { offl[1] = do_l ? a.off : 0; offc[1] = do_c ? b0.off : 0; }
        else   { offl[0] = do_l ? a.off : 0; offc[0] = do_c ? b0.off : 0; }
    }
}

// ---- 4 outputs of an NT-tap FIR over a packed int16 row: out[o] = sum_k taps[k] * s[o + k].
// d[j] = (s[2j], s[2j+1]); even outputs use the dwords as they are, odd outputs the dwords shifted by
// one sample (v_alignbit); every dword pair is one v_dot2c_i32_i16 (2 MACs). ----
template <int NT>
__device__ __forceinline__ void fir4(const int d[NT / 2 + 2], const int tp[NT / 2], int out[4])
{
    int e[NT / 2 + 1];
#pragma unroll
    for (int j = 0; j < NT / 2 + 1; ++j) e[j] = (int)__builtin_amdgcn_alignbit((uint32_t)d[j + 1], (uint32_t)d[j], 16);
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        int acc = 0;
#pragma unroll
        for (int m = 0; m < NT / 2; ++m) {
            const int v = (o & 1) ? e[(o >> 1) + m] : d[(o >> 1) + m];
            acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, v), __builtin_bit_cast(short2v, tp[m]), acc, false);
        }
        out[o] = acc;
    }
}

// Filter taps as pairs ((f[2m], f[2m+1]) = int16 halves of one dword, what v_dot2_i32_i16 multiplies), packed at COMPILE time:
// a wave-uniform filter is then one scalar load instead of 8 byte loads, 8 sign extensions and 12 shift / or (the tap
// set-up was a fifth of k_mc2's scalar instructions).  luma6 = the 6 taps 1..6 of the 4x4-block luma filter.
struct PackedTaps { uint32_t luma[17][4], luma4[16][4], luma6[16][4], chroma[32][2]; };
constexpr uint32_t pack_pair(int a, int b) { return ((uint32_t)a & 0xffffu) | ((uint32_t)b << 16); }
constexpr PackedTaps build_packed_taps()
{
    PackedTaps t{};
    for (int f = 0; f < 17; ++f)
        for (int m = 0; m < 4; ++m) t.luma[f][m] = pack_pair(ovt_mc_luma[f][2 * m], ovt_mc_luma[f][2 * m + 1]);
    for (int f = 0; f < 16; ++f)
        for (int m = 0; m < 4; ++m) {
            t.luma4[f][m] = pack_pair(ovt_mc_luma4[f][2 * m], ovt_mc_luma4[f][2 * m + 1]);
            t.luma6[f][m] = m < 3 ? pack_pair(ovt_mc_luma4[f][2 * m + 1], ovt_mc_luma4[f][2 * m + 2]) : 0u;
        }
    for (int f = 0; f < 32; ++f)
        for (int m = 0; m < 2; ++m) t.chroma[f][m] = pack_pair(ovt_mc_chroma[f][2 * m], ovt_mc_chroma[f][2 * m + 1]);
    return t;
}
__device__ const PackedTaps __attribute__((aligned(16))) g_taps = build_packed_taps();

template <int ND>
__device__ __forceinline__ void load_taps(const uint32_t *row, int tp[ND])
{
#pragma unroll
    for (int m = 0; m < ND; ++m) tp[m] = (int)row[m];
}

template <int NT>
__device__ __forceinline__ void load_row(const void *p, int d[NT / 2 + 2])
{
    // segment starts are 8-byte aligned in LDS (strides are multiples of 4 samples)
    const int2 *q = reinterpret_cast<const int2 *>(p);
#pragma unroll
    for (int j = 0; j < (NT / 2 + 2) / 2; ++j) { const int2 v = q[j]; d[2 * j] = v.x; d[2 * j + 1] = v.y; }
}

// same, starting at an arbitrary SAMPLE index s0 of a dword-aligned row (window rows keep their
// aligned-group offset): dword loads + one v_alignbit per dword when s0 is odd
template <int NT>
__device__ __forceinline__ void load_row_at(const uint16_t *row, int s0, int d[NT / 2 + 2])
{
    const int *q = reinterpret_cast<const int *>(row) + (s0 >> 1);
    int D[NT / 2 + 2];
#pragma unroll
    for (int j = 0; j < NT / 2 + 2; ++j) D[j] = q[j];
    if (s0 & 1) {
#pragma unroll
        for (int j = 0; j < NT / 2 + 1; ++j) d[j] = (int)__builtin_amdgcn_alignbit((uint32_t)D[j + 1], (uint32_t)D[j], 16);
        d[NT / 2 + 1] = (int)((uint32_t)D[NT / 2 + 1] >> 16);
    } else {
#pragma unroll
        for (int j = 0; j < NT / 2 + 2; ++j) d[j] = D[j];
    }
}

// ---- stage 3: vertical pass LDS -> registers.  Lane = one column x 4 consecutive rows (group g):
// P[j] = F_v(t)[x][4g + j] >> 6, the 14-bit intermediate of put_vvc_{qpel,epel}_*. ----
template <int NT>
__device__ __forceinline__ void v_pass(const int16_t *s_ht, int htstride, int log2w, int h, const uint32_t *fv_packed, int lane, int P[4])
{
    int tp[NT / 2];
    load_taps<NT / 2>(fv_packed, tp);
    const int w = 1 << log2w;
    const int ngrp = (h + 3) >> 2;
    if (lane < (ngrp << log2w)) {
        const int x = lane & (w - 1), g = lane >> log2w;
        int d[NT / 2 + 2];
        load_row<NT>(s_ht + x * htstride + 4 * g, d);
        fir4<NT>(d, tp, P);
#pragma unroll
        for (int o = 0; o < 4; ++o) P[o] >>= 6;
    }
}



// ---- NOUT (1, 2 or 4) outputs of an NT-tap FIR over packed int16 dwords d[0 .. (NT + NOUT) / 2] (see fir4) ----
template <int NT, int NOUT>
__device__ __forceinline__ void firn(const int *d, const int tp[NT / 2], int out[NOUT])
{
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        int acc = 0;
#pragma unroll
        for (int m = 0; m < NT / 2; ++m) {
            const int j = (o >> 1) + m;
            const int v = (o & 1) ? (int)__builtin_amdgcn_alignbit((uint32_t)d[j + 1], (uint32_t)d[j], 16) : d[j];
            acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, v), __builtin_bit_cast(short2v, tp[m]), acc, false);
        }
        out[o] = acc;
    }
}

// NT + NOUT - 1 samples starting at SAMPLE index s0 of a dword-aligned row, as packed dwords (last one may be half used)
template <int NT, int NOUT>
__device__ __forceinline__ void load_span(const int16_t *row, int s0, int d[(NT + NOUT) / 2])
{
    constexpr int ND = (NT + NOUT) / 2;                // dwords covering NT + NOUT - 1 samples from an even start
    const int *q = reinterpret_cast<const int *>(row) + (s0 >> 1);
    int D[ND + 1];
#pragma unroll
    for (int j = 0; j < ND + 1; ++j) D[j] = q[j];
    if (s0 & 1) {
#pragma unroll
        for (int j = 0; j < ND; ++j) d[j] = (int)__builtin_amdgcn_alignbit((uint32_t)D[j + 1], (uint32_t)D[j], 16);
    } else {
#pragma unroll
        for (int j = 0; j < ND; ++j) d[j] = D[j];
    }
}


// ---- merged-pass building blocks (k_mc2, k_mcx): one horizontal task = one window row x 4 outputs; one
// vertical call = NOUT consecutive outputs of one transposed-tile column ----
template <int NT>
__device__ __forceinline__ void h_task(const uint16_t *wrow, int off, int x0, const int tp[NT / 2], bool ident, int16_t *ht,
                                       int htstride, int r)
{
    int d[NT / 2 + 2], out[4];
    if (ident) {
        const uint16_t *sp = wrow + off + x0 + NT / 2 - 1;
#pragma unroll
        for (int o = 0; o < 4; ++o) out[o] = (int)sp[o] << 6;
    } else {
        load_row_at<NT>(wrow, off + x0, d);
        fir4<NT>(d, tp, out);
    }
#pragma unroll
    for (int o = 0; o < 4; ++o)
        ht[(x0 + o) * htstride + r] = (int16_t)(out[o] >> (OV_BD - 8));      // a 2-wide chroma block also fills columns 2, 3: never read
}

template <int NT, int NOUT>
__device__ __forceinline__ void v_outputs(const int16_t *col, int s0, const int tp[NT / 2], int P[NOUT])
{
    int d[(NT + NOUT) / 2];
    load_span<NT, NOUT>(col, s0, d);
    firn<NT, NOUT>(d, tp, P);
#pragma unroll
    for (int o = 0; o < NOUT; ++o) P[o] >>= 6;
}


} // namespace
