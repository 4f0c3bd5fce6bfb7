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
#include <stdlib.h>
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

// ---- stage 1: reference window -> LDS.  Fast path (window inside the picture): every lane loads one
// ALIGNED 8-byte group of 4 samples -- QW lanes per window row, 64/QW rows per instruction: 3
// instructions for a 23x23 luma window, 1 for an 11x11 chroma window -- and parks it with one
// ds_write_b64; the sub-group offset `off` (0..3 samples) is resolved by the horizontal pass.
// Slow path (window crosses the picture border): per-sample loads with clamped coordinates
// = emulate_block_border() (rcn_inter.c:148-225), parked at off = 0. ----
template <int QW, int NIT, int COLS, int NITS>
struct WinStage {
    uint2 q[NIT];
    bool fast;
    int off;

    __device__ __forceinline__ void issue(const uint16_t *__restrict__ ref, int rstride, int rw, int rh, int sx0, int sy0,
                                          int ww, int wh, int lane, uint16_t *s_win, int wstride)
    {
        const int ax = sx0 & ~3;
        off = sx0 - ax;
        const int nq = (off + ww + 3) >> 2;
        fast = ax >= 0 && ax + 4 * nq <= rw && sy0 >= 0 && sy0 + wh <= rh && !(rstride & 3);
        if (fast) {
            const int c = lane & (QW - 1), r0 = lane / QW;
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                const int r = (64 / QW) * k + r0;
                if (c < nq && r < wh) q[k] = *reinterpret_cast<const uint2 *>(ref + (sy0 + r) * rstride + ax + 4 * c);
            }
        } else {
            // rare: load + park immediately (keeps the register footprint of the fast path small)
            off = 0;
            const int c = lane & (COLS - 1), r0 = lane / COLS;
            const int sx = ov_clip3(sx0 + c, 0, rw - 1);
#pragma unroll 1
            for (int k = 0; k < NITS; ++k) {
                const int r = (64 / COLS) * k + r0;
                if (c < ww && r < wh) s_win[r * wstride + c] = ref[ov_clip3(sy0 + r, 0, rh - 1) * rstride + sx];
            }
        }
    }

    __device__ __forceinline__ void park(uint16_t *s_win, int wstride, int ww, int wh, int lane) const
    {
        if (fast) {
            const int nq = (off + ww + 3) >> 2;
            const int c = lane & (QW - 1), r0 = lane / QW;
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                const int r = (64 / QW) * k + r0;
                if (c < nq && r < wh) *reinterpret_cast<uint2 *>(s_win + r * wstride + 4 * c) = q[k];
            }
        }
    }
};
typedef WinStage<8, 3, 32, 12> LumaStage;
typedef WinStage<4, 1, 16, 3> ChromaStage;

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

template <int NT>
__device__ __forceinline__ void pack_taps(const int8_t *f, int tp[NT / 2])
{
#pragma unroll
    for (int m = 0; m < NT / 2; ++m) tp[m] = ((int)f[2 * m] & 0xffff) | ((int)f[2 * m + 1] << 16);
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

// ---- stage 2: horizontal pass LDS -> LDS (transposed).  One task = one window row x 4 consecutive
// outputs.  t = F_h(src) >> (BITDEPTH - 8), stored column-major so that stage 3 reads rows again. ----
template <int NT>
__device__ __forceinline__ void h_pass(const uint16_t *s_win, int wstride, int off, int16_t *s_ht, int htstride, int log2w,
                                       int wh, const int8_t *fh, int lane)
{
    int tp[NT / 2];
    pack_taps<NT>(fh, tp);
    const int w = 1 << log2w;
    const int log2seg = log2w > 2 ? log2w - 2 : 0;          // 4-sample segments per row
    const int nout = w < 4 ? w : 4;
    const bool ident = fh[NT / 2 - 1] == 64;                // integer position: t = s << 4, no FIR
    for (int t = lane; t < (wh << log2seg); t += 64) {
        const int r = t >> log2seg, x0 = (t & ((1 << log2seg) - 1)) << 2;
        int d[NT / 2 + 2], out[4];
        if (ident) {
            const uint16_t *sp = s_win + r * wstride + off + x0 + NT / 2 - 1;
#pragma unroll
            for (int o = 0; o < 4; ++o) out[o] = (int)sp[o] << 6;
        } else {
            load_row_at<NT>(s_win + r * wstride, off + x0, d);
            fir4<NT>(d, tp, out);
        }
#pragma unroll
        for (int o = 0; o < 4; ++o)
            if (o < nout) s_ht[(x0 + o) * htstride + r] = (int16_t)(out[o] >> (OV_BD - 8));
    }
}

// ---- stage 3: vertical pass LDS -> registers.  Lane = one column x 4 consecutive rows (group g):
// P[j] = F_v(t)[x][4g + j] >> 6, the 14-bit intermediate of put_vvc_{qpel,epel}_*. ----
template <int NT>
__device__ __forceinline__ void v_pass(const int16_t *s_ht, int htstride, int log2w, int h, const int8_t *fv, int lane, int P[4])
{
    int tp[NT / 2];
    pack_taps<NT>(fv, tp);
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

__device__ __forceinline__ int mc_combine(const ovhip_mc_unit &u, int p0, int p1)
{
    if (u.dir != 3)                  return ov_clip_bd(((u.dir == 1 ? p0 : p1) + 8) >> 4);
    if (u.w0 == 4 && u.w1 == 4)      return ov_clip_bd((p0 + p1 + 16) >> 5);
    return ov_clip_bd((p1 * u.w1 + p0 * u.w0 + 64) >> 7);
}

__global__ __launch_bounds__(64) void k_mc(ovhip_pic dst, RefTable refs, const ovhip_mc_unit *__restrict__ units,
                                            uint32_t n_units, const uint16_t *__restrict__ lmcs_fwd, int ablate)
{
    __shared__ __attribute__((aligned(16))) uint16_t s_wl[2][LUMA_WIN];          // luma windows, list 0 / 1
    __shared__ __attribute__((aligned(16))) uint16_t s_wc[2][2][CHR_WIN];        // chroma windows [Cb/Cr][list]
    __shared__ __attribute__((aligned(16))) int16_t  s_hl[2][16 * HT_STRIDE];    // transposed H-pass tiles
    __shared__ __attribute__((aligned(16))) int16_t  s_hc[2][2][8 * CHT_STRIDE];

    const int lane = threadIdx.x;
    // grid-stride over units: the chip launches ~400 workgroups/us, so one workgroup per unit would be
    // dispatch-bound; a resident grid of single-wave workgroups walks the unit list instead
    for (uint32_t bid = blockIdx.x; bid < n_units; bid += gridDim.x) {
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
                    int v = mc_combine(u, P[0][j], P[1][j]);
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
                    if (4 * g + j < hc) d[j * dst.stride_c] = (uint16_t)mc_combine(u, P[0][j], P[1][j]);
            }
        }
    }
    __syncthreads();          // LDS tiles are reused by the next unit
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
    static int cfg_grid = -1, cfg_ablate = 0;
    if (cfg_grid < 0) {                       // developer knobs (profiling experiments only)
        const char *g = getenv("OVHIP_MC_GRID"), *a = getenv("OVHIP_MC_ABLATE");
        cfg_grid = g ? atoi(g) : 0;
        cfg_ablate = a ? atoi(a) : 0;
    }
    uint32_t grid = cfg_grid > 0 ? (uint32_t)cfg_grid : n_units;
    if (grid > n_units) grid = n_units;
    hipLaunchKernelGGL(k_mc, dim3(grid), dim3(64), 0, ctx->stream, *dst, t, d_units, n_units, d_lmcs_fwd_lut, cfg_ablate);
    OV_LAUNCH_CHECK(ctx, "k_mc");
    return OVHIP_OK;
}

extern "C" int ovhip_mcx_launch(ovhip_ctx *ctx, const ovhip_pic *dst, const ovhip_pic *refs, uint32_t n_refs,
                                const ovhip_mc_unit *d_units, uint32_t n_units, const uint16_t *d_lmcs_fwd_lut,
                                int32_t *d_mv_out)
{
    return ov_fail(ctx, OVHIP_EINVAL, "ovhip_mcx_launch: not built", hipSuccess);
}
