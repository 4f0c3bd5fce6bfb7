// kernels_itx.hip -- K1..K4 on gfx950: inverse quantisation, LFNST, separable inverse
// transform (DCT-II 2..64, DST-VII / DCT-VIII 4..32), residual add (+JCCR, +LMCS chroma scale).
//
// One workgroup per transform block (one wave up to 16x16, four above).  The coefficient tile is
// de-scanned and de-quantised straight from the compact coefficient arena into LDS (one lane
// per 4x4 sub-block, 32 contiguous bytes per lane -> coalesced), the two transform cores the
// block needs are staged in LDS, and both 1-D passes run out of LDS.  Everything a pass multiplies
// is laid out with the summation index k contiguous -- coefficient tile TRANSPOSED, cores as
// int16 [output][k] -- so that one ds_read_b64 of each operand feeds two v_dot2_i32_i16 (4 MACs,
// int32 wrap-around accumulation as in the reference).  No MFMA: exact clip16 rounding between
// the passes, K <= 32.
//
// Replaces, per block, the reference call chain
//   dequant_tb_4x4 -> [compute_lfnst_*] -> tr.func[v] -> tr.func[h] -> ict.add / ict.ict
// (libovvc/rcn_transform_tree.c:415-506, :553-628; rcn_dequant.c:160-312; rcn_lfnst.c:41-162;
//  rcn_transform.c:44-598; rcn_residuals.c:46-222).  Every butterfly in rcn_transform.c is an
// exact integer refactoring of the plain matrix product computed here.
#include "ovvc_common.hip.h"
#include <stdlib.h>
#define OVT_ATTR __device__
#include "vvc_tables.h"

// Occupancy hint (waves per SIMD the register allocator must leave room for; 0 = compiler default).
// -DOV_WPE_ITX=n overrides it for sweeps.
#ifndef OV_WPE_ITX
#define OV_WPE_ITX 8      /* measured (tools/sweep_occupancy.sh, two pictures in flight): itx 61.8 -> 59.0 us */
#endif
#if OV_WPE_ITX > 0
#define OV_OCC_ITX __attribute__((amdgpu_waves_per_eu(OV_WPE_ITX)))
#else
#define OV_OCC_ITX
#endif

namespace {

// ---- transform cores as the passes read them: int16, transposed, T[j * KS + k] = M[k * n + j] for the K = min(n, 32)
// live input rows, zero up to the row length KS = max(K, 8).  Derived from the int8 tables at compile time. ----
constexpr int core_ks(int log2n)   { const int n = 1 << log2n, K = n < 32 ? n : 32; return K < 8 ? 8 : K; }
constexpr int core_size(int log2n) { return (1 << log2n) * core_ks(log2n); }
constexpr int core_off(int type, int log2n)          // type: 0 DST-VII, 1 DCT-VIII (4..32), 2 DCT-II (2..64)
{
    int off = 0;
    for (int t = 0; t < 3; ++t)
        for (int l = (t == 2 ? 1 : 2); l <= (t == 2 ? 6 : 5); ++l) {
            if (t == type && l == log2n) return off;
            off += core_size(l);
        }
    return off;
}
constexpr int CORE_TOTAL = core_off(3, 0);
struct CoreT16 { int16_t v[CORE_TOTAL]; };
constexpr CoreT16 build_cores()
{
    CoreT16 t{};
    const int8_t *const tabs[3][7] = {
        { nullptr, nullptr, ovt_dst7_4, ovt_dst7_8, ovt_dst7_16, ovt_dst7_32, nullptr },
        { nullptr, nullptr, ovt_dct8_4, ovt_dct8_8, ovt_dct8_16, ovt_dct8_32, nullptr },
        { nullptr, ovt_dct2_2, ovt_dct2_4, ovt_dct2_8, ovt_dct2_16, ovt_dct2_32, ovt_dct2_64 } };
    for (int ty = 0; ty < 3; ++ty)
        for (int l = 1; l <= 6; ++l) {
            if (!tabs[ty][l]) continue;
            const int n = 1 << l, K = n < 32 ? n : 32, KS = core_ks(l), off = core_off(ty, l);
            for (int j = 0; j < n; ++j)
                for (int k = 0; k < K; ++k) t.v[off + j * KS + k] = tabs[ty][l][k * n + j];
        }
    return t;
}
__device__ const CoreT16 __attribute__((aligned(16))) g_cores = build_cores();

// offset of a core in g_cores, in units of 16 values, relative to its type's first core: one byte per log2n, so that the
// lookup is a shift and a mask (a 14-way switch on wave-uniform values is ~50 scalar instructions per core)
constexpr uint64_t core_lut(int type)
{
    uint64_t v = 0;
    for (int l = (type == 2 ? 1 : 2); l <= (type == 2 ? 6 : 5); ++l)
        v |= (uint64_t)((core_off(type, l) - core_off(type, type == 2 ? 1 : 2)) / 16) << (8 * l);
    return v;
}
static_assert(core_lut(0) == core_lut(1) && (core_off(2, 6) - core_off(2, 1)) / 16 < 256 && core_off(1, 2) % 16 == 0 && core_off(2, 1) % 16 == 0,
              "core offsets must fit the byte table");
__device__ __forceinline__ const int16_t *tr_core(int type, int log2n)
{
    const uint64_t lut = type == 2 ? core_lut(2) : core_lut(0);
    const int base = type == 2 ? core_off(2, 1) : type == 1 ? core_off(1, 2) : 0;
    return g_cores.v + base + 16 * (int)((lut >> (8 * log2n)) & 0xff);
}
// LDS row length of a k-contiguous tile whose rows hold K values: the padding keeps the ds_read_b64 of 16 rows apart
__device__ __forceinline__ int tile_stride(int K) { return K < 8 ? 8 : K + (K >= 16 ? 4 : 0); }
constexpr int tile_stride_c(int K) { return K < 8 ? 8 : K + (K >= 16 ? 4 : 0); }

__device__ __forceinline__ int dequant1(int c, int scale, int shift, int neg)
{
    // rcn_dequant.c:160-312 -- int32 wrap-around arithmetic, then ov_clip_intp2(v, 16), which is
    // the SYMMETRIC range [-32767, 32767] (ovutils.h:78-92)
    if (neg) return ov_clip3((int)((uint32_t)c * (uint32_t)(scale << shift)), -32767, 32767);
    return ov_clip3(((int)((uint32_t)c * (uint32_t)scale) + ((1 << shift) >> 1)) >> shift, -32767, 32767);
}

__device__ __forceinline__ int nb_rows_of(uint64_t map)   // derive_nb_rows, rcn_transform_tree.c:78-92
{
    uint32_t m = (uint32_t)map | (uint32_t)(map >> 32);
    m |= m >> 16; m |= m >> 8;
    m = (m & 0xff) | 1;
    return (32 - __clz((int)m)) << 2;
}
__device__ __forceinline__ int nb_cols_of(uint64_t map)   // derive_nb_cols, :94-101
{
    return (8 - (__clzll((long long)(map | 1)) >> 3)) << 2;
}

// the eight variants of rcn_residuals.c:46-222 on one sample
__device__ __forceinline__ int residual1(int pix, int r, int mode, int scale)
{
    // r, -r, r >> 1, (-r) >> 1 without a branch (mode is wave-uniform: a switch here is scalar compares and branches per
    // sample): bit 0 negates, bit 1 halves
    const int neg = -(mode & 1);
    int v = ((r ^ neg) - neg) >> ((mode >> 1) & 1);
    // block of an ordered task: the residual itself goes out (to the residual picture, see ResDelta), saturated to int16
    if (mode & OVHIP_RES_STORE) return ov_clip16(v) & 0xffff;
    if (mode & OVHIP_RES_SCALE) {
        int sign = v & (1 << 15);
        int a = ov_clip_bd(abs(v));
        a = (a * scale + (1 << 10)) >> 11;
        v = ov_clip3(sign ? -a : a, -(1 << 15), 1 << 15);
    }
    return ov_clip_bd(pix + v);
}

// Where OVHIP_RES_STORE blocks go: the residual picture has the geometry of the picture being decoded, so it is addressed as
// "same position, other allocation": byte distance of each of its planes from the picture's (0 when there is none).
struct ResDelta { long long d[3]; };

struct ResidualSink {
    uint16_t *dst; int stride; int mode;
    uint16_t *dst2; int stride2; int mode2;
    int scale;
};
typedef short short2v __attribute__((ext_vector_type(2)));

// One 1-D pass out of LDS:  out[i][j] = clip16((sum_{k < kmax} src[i * sstride + k] * core[j * cstride + k] + rnd) >> shift)
// for i < lines, j < n; both operands are k-contiguous (kmax is rounded up to 4: the operands are zero there).
// Lane = one output.  FINAL = false: store int16 to dst[j * dstride + i] (pass 1: the transposed tile pass 2
// reads).  FINAL = true: fuse K4, the residual add into the frame (pass 2: lanes j -> contiguous frame addresses).
template <bool FINAL, int NT, int NK4>
__device__ __forceinline__ void tr_pass_lds_n(const int16_t *src, int sstride, const int16_t *core, int cstride, int log2n,
                                               int kmax, int lines, int shift, int16_t *dst, int dstride, int lane,
                                               const ResidualSink &sink)
{
    const int n = 1 << log2n;
    const int ntask = lines << log2n;
    const int rnd = 1 << (shift - 1);
    const int nk4 = (kmax + 3) >> 2;
    for (int t = lane; t < ntask; t += NT) {
        const int j = t & (n - 1), i = t >> log2n;
        int old = 0, old2 = 0;
        if (FINAL) {
            // the frame reads of the read-modify-write go out BEFORE the MAC loop
            old = sink.dst[i * sink.stride + j];
            if (sink.dst2) old2 = sink.dst2[i * sink.stride2 + j];
        }
        const int2 *sp = reinterpret_cast<const int2 *>(src + i * sstride);
        const int2 *cp = reinterpret_cast<const int2 *>(core + j * cstride);
        int acc = 0;
        if (NK4 > 0) {
            // the common lengths with the loop gone: all operand reads in flight, then the dot products (the loop's counter, compare
            // and branch were as many scalar instructions as the loop had vector ones)
            int2 a[NK4 > 0 ? NK4 : 1], m[NK4 > 0 ? NK4 : 1];
#pragma unroll
            for (int k4 = 0; k4 < NK4; ++k4) { a[k4] = sp[k4]; m[k4] = cp[k4]; }
#pragma unroll
            for (int k4 = 0; k4 < NK4; ++k4) {
                acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, a[k4].x), __builtin_bit_cast(short2v, m[k4].x), acc, false);
                acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, a[k4].y), __builtin_bit_cast(short2v, m[k4].y), acc, false);
            }
        } else
        for (int k4 = 0; k4 < nk4; ++k4) {
            const int2 a = sp[k4], m = cp[k4];
            acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, a.x), __builtin_bit_cast(short2v, m.x), acc, false);
            acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, a.y), __builtin_bit_cast(short2v, m.y), acc, false);
        }
        const int r = ov_clip16((acc + rnd) >> shift);
        if (!FINAL) {
            dst[j * dstride + i] = (int16_t)r;
        } else {
            sink.dst[i * sink.stride + j] = (uint16_t)residual1(old, r, sink.mode, sink.scale);
            if (sink.dst2) sink.dst2[i * sink.stride2 + j] = (uint16_t)residual1(old2, r, sink.mode2, sink.scale);
        }
    }
}

template <bool FINAL, int NT>
__device__ __forceinline__ void tr_pass_lds(const int16_t *src, int sstride, const int16_t *core, int cstride, int log2n,
                                             int kmax, int lines, int shift, int16_t *dst, int dstride, int lane,
                                             const ResidualSink &sink)
{
    // kmax is wave-uniform (a field of the command / the block's significance map)
    const int nk4 = (kmax + 3) >> 2;
    if (nk4 == 1)      tr_pass_lds_n<FINAL, NT, 1>(src, sstride, core, cstride, log2n, kmax, lines, shift, dst, dstride, lane, sink);
    else if (nk4 == 2) tr_pass_lds_n<FINAL, NT, 2>(src, sstride, core, cstride, log2n, kmax, lines, shift, dst, dstride, lane, sink);
    else if (nk4 == 4) tr_pass_lds_n<FINAL, NT, 4>(src, sstride, core, cstride, log2n, kmax, lines, shift, dst, dstride, lane, sink);
    else               tr_pass_lds_n<FINAL, NT, 0>(src, sstride, core, cstride, log2n, kmax, lines, shift, dst, dstride, lane, sink);
}

// Rider of the chroma launch: inverse LMCS mapping of the luma plane (rcn_lmcs_reshape_backward, rcn_lmcs.c:219-231) by
// workgroups n_cmds .. n_cmds + n_extra - 1.  Luma is final once the luma commands and k_lmcs_scale have run and the
// chroma commands never touch it, so the two share a launch instead of paying a kernel boundary each.
template <int NT>
__device__ __forceinline__ void lmcs_inverse_rows(const ovhip_pic &pic, const uint16_t *__restrict__ lut, uint32_t e, uint32_t n_extra,
                                                  uint16_t *s_lut)
{
    for (int i = threadIdx.x; i < 512; i += NT) reinterpret_cast<uint32_t *>(s_lut)[i] = reinterpret_cast<const uint32_t *>(lut)[i];
    __syncthreads();
    const int nvx = pic.w >> 3, tail = pic.w & 7;             // full 8-sample vectors per row (plane 16-byte aligned, stride % 8 == 0)
    for (int y = (int)e; y < pic.h; y += (int)n_extra) {
        uint16_t *row = pic.y + (size_t)y * pic.stride_y;
        for (int v = threadIdx.x; v < nvx; v += NT) {
            uint4 q = *reinterpret_cast<uint4 *>(row + 8 * v);
            uint32_t *d = reinterpret_cast<uint32_t *>(&q);
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] = s_lut[d[k] & 1023] | ((uint32_t)s_lut[(d[k] >> 16) & 1023] << 16);
            *reinterpret_cast<uint4 *>(row + 8 * v) = q;
        }
        if ((int)threadIdx.x < tail) row[8 * nvx + threadIdx.x] = s_lut[row[8 * nvx + threadIdx.x] & 1023];
    }
}

// LDS of one block, int16 entries, carved at run time from the block's own dimensions: s_coef (transform blocks
// [column][row], stride tile_stride(ch); everything else raster [row][column], stride tb_w), s_tmp (pass 1 -> pass 2:
// [row][column k]), s_mv and s_mh (cores [output][k]).  A 64x64 block needs 32*36 + 3 * 64*36 = 8064 entries; a block
// of at most 256 samples with no side above 32 (16x16, 32x8, 8x32, 32x4, ...) at most 1760: its slice is 2048.
#define ITX_LDS_BIG   8064
#define ITX_LDS_SLICE 2048

// One transform block by NT threads (lane 0 .. NT-1): <256> any block up to 64x64, four waves per block (a 64x64
// block is ~200k MACs: one wave alone would be the tail of the launch); <64> blocks of at most 256 samples, one wave and
// a 4 KB slice of LDS.  EVERY path runs the same four barriers, whatever the block needs (LFNST, BDPCM, nothing): that is
// what lets four waves with four different small blocks share a 256-thread workgroup (k_itx_all); a wave without a
// block (valid = false) only keeps the barriers company.
// where a block's residual goes: the plane(s) of the command, or the residual picture for blocks of ordered tasks
__device__ __forceinline__ ResidualSink make_sink(const ovhip_pic &pic, const ResDelta &rd, const ovhip_tb_cmd &c, const int16_t *__restrict__ lmcs_scales)
{
    ResidualSink sink;
    sink.dst = ov_plane(pic, c.plane, sink.stride) + c.y * sink.stride + c.x;
    sink.mode = c.res_mode;
    sink.dst2 = nullptr; sink.stride2 = 0; sink.mode2 = c.res_mode2;
    if (c.plane2 != 0xff) sink.dst2 = ov_plane(pic, c.plane2, sink.stride2) + c.y * sink.stride2 + c.x;
    if (c.res_mode & OVHIP_RES_STORE) {
        const long long d1 = c.plane == 0 ? rd.d[0] : (c.plane == 1 ? rd.d[1] : rd.d[2]);
        sink.dst = reinterpret_cast<uint16_t *>(reinterpret_cast<char *>(sink.dst) + d1);
        if (sink.dst2) {
            const long long d2 = c.plane2 == 0 ? rd.d[0] : (c.plane2 == 1 ? rd.d[1] : rd.d[2]);
            sink.dst2 = reinterpret_cast<uint16_t *>(reinterpret_cast<char *>(sink.dst2) + d2);
        }
    }
    sink.scale = (c.res_mode & OVHIP_RES_SCALE_IDX) ? lmcs_scales[c.c_scale] : c.c_scale;   // device-derived chroma scale (K11)
    return sink;
}

// A one-wave block (NT == 64) exchanges data between its lanes through its own slice of LDS only: program order of the wave's DS
// instructions + a compiler fence is a barrier for it (as in kernels_intra.hip); four-wave blocks need the workgroup barrier.
template <int NT>
__device__ __forceinline__ void block_sync()
{
    if (NT == 64) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else __syncthreads();
}

// LW / LH >= 0: the block's shape as a compile-time constant (k_itx_all dispatches the common shapes of the one-wave class on the
// wave-uniform command: tile carve-up, strides, chunk counts and every loop bound below then fold -- the generic body executed more
// scalar than vector instructions, profiles/r03d_pmc_sq.txt).  PLAIN: the command is a transform block off the arena's 4x4
// sub-blocks without LFNST (what nearly every block is): the raster / transform-skip / BDPCM / LFNST / DC arms are not compiled in.
template <int NT, int LW = -1, int LH = -1, bool PLAIN = false>
__device__ __forceinline__ void itx_block(const ovhip_pic &pic, const ResDelta &rd, const ovhip_tb_cmd &c, bool valid, const int16_t *__restrict__ arena,
                                          const int16_t *__restrict__ lmcs_scales, int ablate, int lane, int16_t *lds)
{
    const int log2_w = LW >= 0 ? LW : c.log2_w, log2_h = LH >= 0 ? LH : c.log2_h;
    const int tb_w = 1 << log2_w, tb_h = 1 << log2_h;
    const int kind = PLAIN ? (int)OVHIP_TB_TR : (c.kind & 0x3f);
    const bool raster = PLAIN ? false : (bool)(c.kind & OVHIP_TB_FLAG_RASTER), bdpcm = PLAIN ? false : (bool)(c.kind & OVHIP_TB_FLAG_BDPCM);
    if (PLAIN) ablate = 0;
    const int cw = min(tb_w, 32), ch = min(tb_h, 32);
    const int16_t *src = arena + c.coef_off;

    // ---- DC-only blocks (a quarter of the blocks of an inter picture): inverse_dct_ii_dc (rcn_transform.c:576-598) needs the
    //      one coefficient -- no tiles, no cores, no LDS.  One-wave blocks only: four-wave blocks meet at workgroup barriers ----
    if (NT == 64 && kind == OVHIP_TB_DC && !ablate) {
        if (!valid) return;
        const bool have = raster || (c.sig_sb_map & 1);              // (an empty first sub-block is staged as zeros, not de-quantised)
        const int lvl = have ? (int)src[0] : 0;
        const int c0 = (!have || bdpcm) ? lvl : dequant1(lvl, c.dq_scale, c.dq_shift, c.dq_neg);
        const int flat_val = ov_clip16(((((int)(int16_t)c0 + 1) >> 1) + (1 << (14 - OV_BD - 1))) >> (14 - OV_BD));
        const ResidualSink sink = make_sink(pic, rd, c, lmcs_scales);
        for (int i0 = lane; i0 < tb_w * tb_h; i0 += 4 * NT) {
            int old[4], old2[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + NT * q;
                if (i < tb_w * tb_h) {
                    const int x = i & (tb_w - 1), y = i >> log2_w;
                    old[q] = sink.dst[y * sink.stride + x];
                    old2[q] = sink.dst2 ? (int)sink.dst2[y * sink.stride2 + x] : 0;
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + NT * q;
                if (i < tb_w * tb_h) {
                    const int x = i & (tb_w - 1), y = i >> log2_w;
                    sink.dst[y * sink.stride + x] = (uint16_t)residual1(old[q], flat_val, sink.mode, sink.scale);
                    if (sink.dst2) sink.dst2[y * sink.stride2 + x] = (uint16_t)residual1(old2[q], flat_val, sink.mode2, sink.scale);
                }
            }
        }
        return;
    }

    // ---- K1 loads first: the lane's 4x4 sub-block of levels (HBM), then the transform cores (L2-resident tables,
    //      only what this block needs), so that both round trips overlap ----
    const int l2nx = max(min(log2_w, 5) - 2, 0), nx = 1 << l2nx, ny = ch >> 2;   // blocks narrower than 4 arrive in raster order
    const int sx = lane & (nx - 1), sy = lane >> l2nx, bit = sy * 8 + sx;
    const bool descan = !raster && !(ablate & 2) && cw >= 4 && lane < nx * ny;
    const bool sig = descan && ((c.sig_sb_map >> bit) & 1);
    int4 v0 = make_int4(0, 0, 0, 0), v1 = v0;
    if (valid && sig) {
        const int rank = __popcll(c.sig_sb_map & ((1ull << bit) - 1));
        const int4 *p = reinterpret_cast<const int4 *>(src + rank * 16);
        v0 = p[0]; v1 = p[1];
    }
    const int kv = min(tb_h, 32), kh = min(tb_w, 32);
    // 1 x N / N x 1 blocks of intra sub-partitions (rcn_1xX_tb, rcn_Xx1_tb, rcn_transform_tree.c:947-962, :1011-1027): ONE
    // transform along the long side with the second pass's shift + 1; staged raster like a transform-skip block
    const bool one_d = kind == OVHIP_TB_TR && (log2_w == 0 || log2_h == 0);
    const bool is_tr = kind == OVHIP_TB_TR && !one_d;
    const int cs = tile_stride(ch);                       // s_coef row length of a transform block
    const int msv = tile_stride(kv), msh = tile_stride(kh);
    // the four tiles (sizes rounded to 8 entries: every tile starts 16-byte aligned)
    const int n_coef = (max(cw * cs, is_tr ? 0 : tb_w * tb_h) + 7) & ~7, n_tmp = (tb_h * msh + 7) & ~7, n_mv = (tb_h * msv + 7) & ~7;
    int16_t *const s_coef = lds, *const s_tmp = lds + n_coef, *const s_mv = s_tmp + n_tmp, *const s_mh = s_mv + n_mv;
    if (valid && is_tr && !(ablate & 1)) {
        // 16-byte chunks (8 values of one core row): at most 64 * 32 / 8 = 256 <= NT per core for <6, 256>, 32 for <4, 64>
        const int l2cv = kv >= 32 ? 2 : kv >= 16 ? 1 : 0, l2ch = kh >= 32 ? 2 : kh >= 16 ? 1 : 0;   // chunks per row
        const int nv = tb_h << l2cv, nh = tb_w << l2ch;
        const uint4 *mv4 = reinterpret_cast<const uint4 *>(tr_core(c.tr_v, log2_h));
        const uint4 *mh4 = reinterpret_cast<const uint4 *>(tr_core(c.tr_h, log2_w));
        // (one pass for NT = 256; a 32-point core of a one-wave block is 128 chunks: two passes)
        for (int i = lane; i < max(nv, nh); i += NT) {
            uint4 a = make_uint4(0, 0, 0, 0), b = a;
            if (i < nv) a = mv4[i];
            if (i < nh) b = mh4[i];
            if (i < nv) {
                uint2 *d = reinterpret_cast<uint2 *>(s_mv + (i >> l2cv) * msv + ((i & ((1 << l2cv) - 1)) << 3));
                d[0] = make_uint2(a.x, a.y); d[1] = make_uint2(a.z, a.w);
            }
            if (i < nh) {
                uint2 *d = reinterpret_cast<uint2 *>(s_mh + (i >> l2ch) * msh + ((i & ((1 << l2ch) - 1)) << 3));
                d[0] = make_uint2(b.x, b.y); d[1] = make_uint2(b.z, b.w);
            }
        }
    }

    // ---- K1: de-scan + de-quantise into LDS raster [ch][cw] ----
    if (!valid || (ablate & 2)) {
    } else if (raster) {
        if (is_tr) {
            // (blocks narrower than 4): transposed, rows padded with zeros to the 4 values a pass step reads
            for (int i = lane; i < tb_w * tb_h; i += NT) {
                const int x = i & (tb_w - 1), y = i >> log2_w;
                // (a 64-point transform reads its first 32 inputs only: rows 32..63 of a 2x64 ISP block, columns 32..63 of a 64x2 one are
                //  not staged)
                if (y < ch && x < cw) s_coef[x * cs + y] = (int16_t)dequant1(src[i], c.dq_scale, c.dq_shift, c.dq_neg);
                if (tb_h < 4 && y == 0 && x < cw) { s_coef[x * cs + 2] = 0; s_coef[x * cs + 3] = 0; }
            }
        } else {
            for (int i = lane; i < tb_w * tb_h; i += NT)
                s_coef[i] = (kind == OVHIP_TB_TS_RAW || bdpcm) ? src[i] : (int16_t)dequant1(src[i], c.dq_scale, c.dq_shift, c.dq_neg);
        }
    } else if (descan) {
        const int w[8] = { v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w };       // zeros for an empty sub-block
        int o[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int word = w[r * 2 + (q >> 1)];
                const int cv = (q & 1) ? (word >> 16) : (int)(int16_t)(word & 0xffff);
                o[r][q] = (bdpcm || !sig) ? cv : dequant1(cv, c.dq_scale, c.dq_shift, c.dq_neg);
            }
        // one 8-byte store per row (raster) or per column (transform blocks: transposed tile); all 8-byte aligned
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            uint2 pk;
            if (is_tr) {
                pk.x = (uint32_t)(o[0][a] & 0xffff) | ((uint32_t)o[1][a] << 16);
                pk.y = (uint32_t)(o[2][a] & 0xffff) | ((uint32_t)o[3][a] << 16);
                *reinterpret_cast<uint2 *>(s_coef + (sx * 4 + a) * cs + sy * 4) = pk;
            } else {
                pk.x = (uint32_t)(o[a][0] & 0xffff) | ((uint32_t)o[a][1] << 16);
                pk.y = (uint32_t)(o[a][2] & 0xffff) | ((uint32_t)o[a][3] << 16);
                *reinterpret_cast<uint2 *>(s_coef + (sy * 4 + a) * cw + sx * 4) = pk;
            }
        }
    }
    block_sync<NT>();                                  // barrier 1: tiles staged

    const ResidualSink sink = make_sink(pic, rd, c, lmcs_scales);

    if (ablate & 4) valid = false;
    const bool tr = valid && is_tr, lf = !PLAIN && tr && (c.lfnst & 1), bd = valid && !is_tr && bdpcm;
    int nb_row = 0, nb_col = 0;
    if (tr) {
        if (raster) {
            const int l2sw = log2_w > 2 ? 3 : 1;
            nb_col = kv;
            nb_row = (nb_rows_of(c.sig_sb_map) >> 2) << l2sw;
        } else {
            nb_row = nb_rows_of(c.sig_sb_map);
            nb_col = nb_cols_of(c.sig_sb_map);     // rows of the tile that can hold non-zero data
        }
    }
    // ---- K2: LFNST on the first sub-block (rcn_lfnst.c:41-162): products first, written back after the barrier ----
    const bool is8 = log2_w >= 3 && log2_h >= 3;
    const int nout = is8 ? 48 : 16;
    int out = 0;
    if (lf) {
        const int set = (c.lfnst >> 1) & 3, idx = (c.lfnst >> 3) & 1;
        const int8_t *m = is8 ? ovt_lfnst_8x8[set][idx] : ovt_lfnst_4x4[set][idx];
        const int nin = is8 ? 16 : (min(log2_w, 5) == min(log2_h, 5) ? 8 : 16);
        if (lane < nout) {
            // diagonal scan of the 4x4 sub-block: constant 0xfbe7ad369c258140 (rcn_lfnst.c:46-53)
            const uint64_t scan = 0xfbe7ad369c258140ull;
            int s = 0;
            for (int j = 0; j < nin; ++j) {
                const int pos = (int)((scan >> (4 * j)) & 0xf);
                s += (int)s_coef[(pos & 3) * cs + (pos >> 2)] * (int)m[lane + j * nout];
            }
            out = ov_clip3((s + 64) >> 7, -(1 << 15), 1 << 15);
        }
    } else if (bd) {
        // ---- block DPCM (rcn_bdpcm_tb, rcn_transform_tree.c:631-688): running sum of the LEVELS along a row / column with
        // int16 saturation (lane = one row / column: the saturation makes the scan order-dependent), then de-quantisation ----
        if (c.tr_h == 0) {
            if (lane < tb_h) {
                int acc = s_coef[lane * tb_w];
                for (int x = 1; x < tb_w; ++x) { acc = ov_clip3(acc + s_coef[lane * tb_w + x], -(1 << 15), (1 << 15) - 1); s_coef[lane * tb_w + x] = (int16_t)acc; }
            }
        } else if (lane < tb_w) {
            int acc = s_coef[lane];
            for (int y = 1; y < tb_h; ++y) { acc = ov_clip3(acc + s_coef[y * tb_w + lane], -(1 << 15), (1 << 15) - 1); s_coef[y * tb_w + lane] = (int16_t)acc; }
        }
    }
    if (!PLAIN) block_sync<NT>();                      // barrier 2 (a PLAIN block is a one-wave block: its barriers are its own)
    if (lf) {
        const int tr_flag = (c.lfnst >> 4) & 1;
        if (lane < nout) {
            int r, q;
            if (!is8)           { r = lane >> 2; q = lane & 3; }
            else if (lane < 32) { r = lane >> 3; q = lane & 7; }
            else                { r = 4 + ((lane - 32) >> 2); q = lane & 3; }
            if (tr_flag) { int t = r; r = q; q = t; }
            s_coef[q * cs + r] = (int16_t)out;
        }
        nb_row = 4 << (int)is8;               // rcn_transform_tree.c:474-475
        nb_col = max(nb_col, nb_row);
    } else if (bd && kind == OVHIP_TB_TS) {
        for (int i = lane; i < tb_w * tb_h; i += NT) s_coef[i] = (int16_t)dequant1(s_coef[i], c.dq_scale, c.dq_shift, c.dq_neg);
    }
    if (!PLAIN) block_sync<NT>();                      // barrier 3
    const int ts = tile_stride(kh);
    if (tr) {
        nb_row = min(nb_row, tb_w);
        const int k1 = min(nb_col, kv);
        // ---- K3: vertical pass (shift 7) of the coefficient columns i < nb_row:  tmp[row j][i] ----
        if ((nb_row & 3) && lane < tb_h)                     // pass 2 reads 4 values per step: zero the tail of a short row
            for (int e = nb_row; e < ((nb_row + 3) & ~3); ++e) s_tmp[lane * ts + e] = 0;
        tr_pass_lds<false, NT>(s_coef, cs, s_mv, msv, log2_h, k1, nb_row, 7, s_tmp, ts, lane, sink);
    }
    if (valid && one_d) {
        const int l2n = log2_w == 0 ? log2_h : log2_w, n = 1 << l2n, kmax = min(n, 32), ks = core_ks(l2n);
        const int16_t *core = tr_core(log2_w == 0 ? c.tr_v : c.tr_h, l2n);
        if (lane < n) {
            int acc = 0;
            for (int k = 0; k < kmax; ++k) acc += (int)s_coef[k] * (int)core[lane * ks + k];
            s_tmp[lane] = (int16_t)ov_clip16((acc + (1 << (20 - OV_BD))) >> (20 - OV_BD + 1));
        }
    }
    block_sync<NT>();                                  // barrier 4
    if (tr) {
        // ---- horizontal pass (shift 20 - bitdepth) fused with K4; columns >= nb_row of tmp are zero: not read ----
        const int k2 = min(nb_row, kh);
        tr_pass_lds<true, NT>(s_tmp, ts, s_mh, msh, log2_w, k2, tb_h, 20 - OV_BD, nullptr, 0, lane, sink);
        return;
    }
    if (!valid) return;

    // ---- DC shortcut / transform skip: K4 directly ----
    const bool flat = kind == OVHIP_TB_DC;
    // inverse_dct_ii_dc, rcn_transform.c:576-598
    const int flat_val = ov_clip16(((((int)s_coef[0] + 1) >> 1) + (1 << (14 - OV_BD - 1))) >> (14 - OV_BD));
    // 4 samples per lane and iteration; all frame reads of an iteration are issued before the first store
    for (int i0 = lane; i0 < tb_w * tb_h; i0 += 4 * NT) {
        int old[4], old2[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + NT * q;
            if (i < tb_w * tb_h) {
                const int x = i & (tb_w - 1), y = i >> log2_w;
                old[q] = sink.dst[y * sink.stride + x];
                old2[q] = sink.dst2 ? (int)sink.dst2[y * sink.stride2 + x] : 0;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + NT * q;
            if (i < tb_w * tb_h) {
                const int x = i & (tb_w - 1), y = i >> log2_w;
                const int r = flat ? flat_val : one_d ? (int)s_tmp[i] : (int)s_coef[y * tb_w + x];   // TS blocks are <= 32 wide: raster stride tb_w
                sink.dst[y * sink.stride + x] = (uint16_t)residual1(old[q], r, sink.mode, sink.scale);
                if (sink.dst2) sink.dst2[y * sink.stride2 + x] = (uint16_t)residual1(old2[q], r, sink.mode2, sink.scale);
            }
        }
    }
}

// One lane per sample, 256 / (w x h) blocks per workgroup: the "tiny" tail of a small class (ovhip_rec_tb_cmds_split_tiny_: plain
// 4x4, 8x4 and 4x8 transform blocks and DC blocks of those shapes -- over a third of a 4K picture's blocks, nearly all of them chroma).
// A wave of its own per such block left 48 / 32 lanes idle and paid a block's scalar work (command decode, sink, core lookup) for
// 16 / 32 samples; here the command is per-lane data, lane (r, q) of a block produces ONE value in each pass, and the two passes go
// through 4 x w x h bytes of LDS per block.  Arithmetic = itx_block's: de-quantise, vertical pass >> 7 with the int16 clip, horizontal
// pass >> (20 - bitdepth), residual1 into the frame (or the residual picture).  (Columns / rows the significance map rules out are
// computed as the zeros they are.)
// dot product of K (4, 8) int16 pairs, both operands k-contiguous and 8 / 16-byte aligned; terms from kmax on are zero by
// construction (kmax a multiple of 4: the significance map works in 4x4 sub-blocks)
template <int K>
__device__ __forceinline__ int itx_dot(const int16_t *a, const int16_t *b, int kmax)
{
    int acc = 0;
    if (K == 4) {
        const uint2 x = *reinterpret_cast<const uint2 *>(a), y = *reinterpret_cast<const uint2 *>(b);
        acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, x.x), __builtin_bit_cast(short2v, y.x), acc, false);
        acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, x.y), __builtin_bit_cast(short2v, y.y), acc, false);
    } else {
#pragma unroll
        for (int k = 0; k < K; k += 8) {
            if (k < kmax) {
                const uint4 x = *reinterpret_cast<const uint4 *>(a + k), y = *reinterpret_cast<const uint4 *>(b + k);
                acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, x.x), __builtin_bit_cast(short2v, y.x), acc, false);
                acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, x.y), __builtin_bit_cast(short2v, y.y), acc, false);
                acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, x.z), __builtin_bit_cast(short2v, y.z), acc, false);
                acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, x.w), __builtin_bit_cast(short2v, y.w), acc, false);
            }
        }
    }
    return acc;
}

template <int LW, int LH>
__device__ __forceinline__ void itx_tiny(const ovhip_pic &pic, const ResDelta &rd, const ovhip_tb_cmd *__restrict__ cmds, uint32_t i, bool valid,
                                         const int16_t *__restrict__ arena, const int16_t *__restrict__ lmcs_scales, int16_t *s /* 2 w h entries */, int l)
{
    constexpr int W = 1 << LW, H = 1 << LH, N = W * H;
    static_assert(N <= 64 && W <= 8 && H <= 8, "a block is at most one wave: its fences are the wave's (tried with 16-point shapes behind workgroup barriers: slower)");
    const ovhip_tb_cmd c = cmds[valid ? i : 0];
    const int r = l >> LW, q = l & (W - 1);
    // the 4x4 sub-block that holds coefficient (r, q): bit sb_y * 8 + sb_x of the map, its rank among the set bits = its place in the arena
    const int bit = (r >> 2) * 8 + (q >> 2);
    const bool sig = (c.sig_sb_map >> bit) & 1;
    const int rank = __popcll(c.sig_sb_map & ((1ull << bit) - 1));
    const int16_t *src = arena + c.coef_off + rank * 16 + (r & 3) * 4 + (q & 3);
    // (both forms are computed by every lane -- a wave holds blocks of both kinds, and the fences stay outside divergent code)
    const int co = sig ? dequant1((int)src[0], c.dq_scale, c.dq_shift, c.dq_neg) : 0;
    s[q * H + r] = (int16_t)co;                                   // [column][row], as pass 1 reads it
    block_sync<64>();
    // (4- and 8-point cores: rows of 8; coefficient rows / columns the significance map rules out are the zeros they are)
    int acc = itx_dot<H>(s + q * H, tr_core(c.tr_v, LH) + r * 8, H);                 // output row r of coefficient column q
    s[N + r * W + q] = (int16_t)ov_clip16((acc + 64) >> 7);
    block_sync<64>();
    acc = itx_dot<W>(s + N + r * W, tr_core(c.tr_h, LW) + q * 8, W);                // output column q of row r
    int res = ov_clip16((acc + (1 << (20 - OV_BD - 1))) >> (20 - OV_BD));
    if (c.kind == OVHIP_TB_DC) {
        // inverse_dct_ii_dc (rcn_transform.c:576-598) on the block's first coefficient (the block's lane 0 staged it at s[0])
        res = ov_clip16(((((int)s[0] + 1) >> 1) + (1 << (14 - OV_BD - 1))) >> (14 - OV_BD));
    }
    if (!valid) return;
    const ResidualSink sink = make_sink(pic, rd, c, lmcs_scales);
    const int old = sink.dst[r * sink.stride + q];
    const int old2 = sink.dst2 ? (int)sink.dst2[r * sink.stride2 + q] : 0;
    sink.dst[r * sink.stride + q] = (uint16_t)residual1(old, res, sink.mode, sink.scale);
    if (sink.dst2) sink.dst2[r * sink.stride2 + q] = (uint16_t)residual1(old2, res, sink.mode2, sink.scale);
}

// the tiny tail of a small class, shape by shape: counts of ovhip_rec_tb_cmds_split_tiny_ (8x8, 4x8, 8x4, 4x4 in list order)
struct TinyCounts { uint32_t n[4]; };

// ONE launch for a sorted command list (ovhip_rec_tb_cmds_split): workgroups [0, n_large) take one block of any size
// each, four waves on it; the next ceil(n_small / 4) take FOUR blocks <= 16x16 each, one per wave, each wave with its own
// 2.5 KB slice of LDS; n_extra more are the inverse-LMCS rider.  (Big and small blocks used to be two launches: the
// kernel boundary and the second launch tail cost more than the barriers the small blocks now share.)
__global__ __launch_bounds__(256) OV_OCC_ITX void k_itx_all(ovhip_pic pic, const ovhip_tb_cmd *__restrict__ cmds, uint32_t n_large,
                                                  uint32_t n_small, const int16_t *__restrict__ arena,
                                                  const int16_t *__restrict__ lmcs_scales, int ablate,
                                                  const uint16_t *__restrict__ lmcs_inv_lut, uint32_t n_extra, ResDelta rd, TinyCounts tc)
{
    __shared__ __attribute__((aligned(16))) int16_t lds[ITX_LDS_BIG > 4 * ITX_LDS_SLICE ? ITX_LDS_BIG : 4 * ITX_LDS_SLICE];
    // the last commands of the n_small are tiny blocks, a lane per sample (workgroups behind the quads): 4x8 and 8x4 blocks eight to a
    // workgroup, 4x4 blocks sixteen
    const uint32_t n_tiny = tc.n[0] + tc.n[1] + tc.n[2] + tc.n[3];
    const uint32_t gq = (tc.n[0] + 3) >> 2, g0 = (tc.n[1] + 7) >> 3, g1 = (tc.n[2] + 7) >> 3, g2 = (tc.n[3] + 15) >> 4, n_tg = gq + g0 + g1 + g2;
    const uint32_t n_wave = n_small - n_tiny;
    const uint32_t b = blockIdx.x, n_quads = (n_wave + 3) >> 2;
    // XCD-aware order (see k_mc2): each XCD takes a contiguous chunk of the sorted list, so blocks that share frame
    // cache lines meet in one L2
    if (b < n_large) {
        itx_block<256>(pic, rd, cmds[ov_xcd_slot(b, n_large)], true, arena, lmcs_scales, ablate, threadIdx.x, lds);
    } else if (b < n_large + n_quads) {
        const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);          // wave-uniform: the command stays in SGPRs
        const uint32_t i = ov_xcd_slot(b - n_large, n_quads) * 4 + w;
        const bool valid = i < n_wave;
        const ovhip_tb_cmd &c = cmds[n_large + (valid ? i : 0)];
        int16_t *const slice = lds + w * ITX_LDS_SLICE;
        const int lane = threadIdx.x & 63;
        // one-wave blocks synchronise inside their wave only (block_sync<64>): each wave may take its own path
        const bool plain = valid && !ablate && (c.kind & 0x3f) == OVHIP_TB_TR && !(c.kind & (OVHIP_TB_FLAG_RASTER | OVHIP_TB_FLAG_BDPCM)) && !(c.lfnst & 1);
        const int shape = plain ? (c.log2_w << 3 | c.log2_h) : -1;
#define ITX_SHAPE(lw, lh) case (lw) << 3 | (lh): itx_block<64, lw, lh, true>(pic, rd, c, true, arena, lmcs_scales, 0, lane, slice); break;
        switch (shape) {
        ITX_SHAPE(2, 2) ITX_SHAPE(3, 2) ITX_SHAPE(2, 3) ITX_SHAPE(3, 3) ITX_SHAPE(4, 2) ITX_SHAPE(2, 4) ITX_SHAPE(4, 3) ITX_SHAPE(3, 4)
        ITX_SHAPE(4, 4) ITX_SHAPE(5, 2) ITX_SHAPE(2, 5) ITX_SHAPE(5, 3) ITX_SHAPE(3, 5)
        default: itx_block<64>(pic, rd, c, valid, arena, lmcs_scales, ablate, lane, slice);
        }
#undef ITX_SHAPE
    } else if (b < n_large + n_quads + n_tg) {
        uint32_t g = b - n_large - n_quads;
        const ovhip_tb_cmd *base = cmds + n_large + n_wave;
        if (g < gq) {
            const uint32_t blk = threadIdx.x >> 6, i = ov_xcd_slot(g, gq) * 4 + blk;
            itx_tiny<3, 3>(pic, rd, base, i, i < tc.n[0], arena, lmcs_scales, lds + 128 * blk, (int)(threadIdx.x & 63));
        } else if ((g -= gq) < g0) {
            const uint32_t blk = threadIdx.x >> 5, i = ov_xcd_slot(g, g0) * 8 + blk;
            itx_tiny<2, 3>(pic, rd, base + tc.n[0], i, i < tc.n[1], arena, lmcs_scales, lds + 64 * blk, (int)(threadIdx.x & 31));
        } else if ((g -= g0) < g1) {
            const uint32_t blk = threadIdx.x >> 5, i = ov_xcd_slot(g, g1) * 8 + blk;
            itx_tiny<3, 2>(pic, rd, base + tc.n[0] + tc.n[1], i, i < tc.n[2], arena, lmcs_scales, lds + 64 * blk, (int)(threadIdx.x & 31));
        } else {
            g -= g1;
            const uint32_t blk = threadIdx.x >> 4, i = ov_xcd_slot(g, g2) * 16 + blk;
            itx_tiny<2, 2>(pic, rd, base + tc.n[0] + tc.n[1] + tc.n[2], i, i < tc.n[3], arena, lmcs_scales, lds + 32 * blk, (int)(threadIdx.x & 15));
        }
    } else {
        lmcs_inverse_rows<256>(pic, lmcs_inv_lut, b - n_large - n_quads - n_tg, n_extra, reinterpret_cast<uint16_t *>(lds));
    }
}

} // namespace

static int itx_ablate()
{
    static int cfg_ablate = -1;
#ifdef OVHIP_TUNING          // experiment knobs read the environment only in tuning builds (make EXTRA=-DOVHIP_TUNING): a stray variable
    if (cfg_ablate < 0) { const char *a = getenv("OVHIP_ITX_ABLATE"); cfg_ablate = a ? atoi(a) : 0; }   // must not change a decoder's results
#else
    cfg_ablate = 0;
#endif
    return cfg_ablate;
}

static int itx_launch(ovhip_ctx *ctx, const ovhip_pic *dst, const ovhip_tb_cmd *d_cmds, uint32_t n_large, uint32_t n_small,
                      const int16_t *d_coefs, const int16_t *d_lmcs_scales, const uint16_t *d_bwd_lut, const ovhip_pic *res = nullptr,
                      const uint32_t *tiny3 = nullptr)
{
    TinyCounts tc = { { tiny3 ? tiny3[0] : 0u, tiny3 ? tiny3[1] : 0u, tiny3 ? tiny3[2] : 0u, tiny3 ? tiny3[3] : 0u } };
    if ((uint64_t)tc.n[0] + tc.n[1] + tc.n[2] + tc.n[3] > n_small) return ov_fail(ctx, OVHIP_EINVAL, "itx launch: more tiny blocks than small ones", hipSuccess);
    ResDelta rd = { { 0, 0, 0 } };
    if (res) {
        if (res->w != dst->w || res->h != dst->h || res->stride_y != dst->stride_y || res->stride_c != dst->stride_c)
            return ov_fail(ctx, OVHIP_EINVAL, "itx launch: the residual picture must have the picture's geometry", hipSuccess);
        rd.d[0] = (char *)res->y - (char *)dst->y; rd.d[1] = (char *)res->cb - (char *)dst->cb; rd.d[2] = (char *)res->cr - (char *)dst->cr;
    }
    // one workgroup per big block / per four small blocks measured faster than a resident grid-stride grid (79 vs 119 us
    // at 4K; and again with the next block's loads software-pipelined: the kernel is issue-bound, not latency-bound)
    const uint32_t n_extra = d_bwd_lut ? (uint32_t)(dst->h + 3) / 4 : 0;      // the inverse-LMCS rider: four luma rows per workgroup
    if (itx_ablate()) tc.n[0] = tc.n[1] = tc.n[2] = tc.n[3] = 0;    // (the ablation switches are the wave-per-block body's)
    const uint32_t n_tiny = tc.n[0] + tc.n[1] + tc.n[2] + tc.n[3];
    const uint32_t grid = n_large + (n_small - n_tiny + 3) / 4 + (tc.n[0] + 3) / 4 + (tc.n[1] + 7) / 8 + (tc.n[2] + 7) / 8 + (tc.n[3] + 15) / 16 + n_extra;
    if (!grid) return OVHIP_OK;
    hipLaunchKernelGGL(k_itx_all, dim3(grid), dim3(256), 0, ctx->stream, *dst, d_cmds, n_large, n_small, d_coefs,
                       d_lmcs_scales, itx_ablate(), d_bwd_lut, n_extra, rd, tc);
    OV_LAUNCH_CHECK(ctx, "k_itx_all");
    return OVHIP_OK;
}

extern "C" int ovhip_itx_launch_classes(ovhip_ctx *ctx, const ovhip_pic *dst, const ovhip_tb_cmd *d_cmds,
                                        uint32_t n_large, uint32_t n_small, const int16_t *d_coefs,
                                        const int16_t *d_lmcs_scales)
{
    if (!ctx || !dst) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    if (!n_large && !n_small) return OVHIP_OK;
    if (!d_cmds || !d_coefs) return ov_fail(ctx, OVHIP_EINVAL, "ovhip_itx_launch: null buffer", hipSuccess);
    return itx_launch(ctx, dst, d_cmds, n_large, n_small, d_coefs, d_lmcs_scales, nullptr);
}

extern "C" int ovhip_itx_launch_classes_res(ovhip_ctx *ctx, const ovhip_pic *dst, const ovhip_pic *res, const ovhip_tb_cmd *d_cmds,
                                            uint32_t n_large, uint32_t n_small, const int16_t *d_coefs,
                                            const int16_t *d_lmcs_scales)
{
    if (!ctx || !dst || !res) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    if ((n_large + n_small) && (!d_cmds || !d_coefs))
        return ov_fail(ctx, OVHIP_EINVAL, "ovhip_itx_launch_classes_res: null commands / coefficients", hipSuccess);
    return itx_launch(ctx, dst, d_cmds, n_large, n_small, d_coefs, d_lmcs_scales, nullptr, res);
}

extern "C" int ovhip_itx_launch_chroma_lmcs(ovhip_ctx *ctx, const ovhip_pic *dst, const ovhip_tb_cmd *d_cmds,
                                            uint32_t n_large, uint32_t n_small, const int16_t *d_coefs,
                                            const int16_t *d_lmcs_scales, const uint16_t *d_bwd_lut)
{
    if (!ctx || !dst || !d_bwd_lut) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    if ((n_large || n_small) && (!d_cmds || !d_coefs)) return ov_fail(ctx, OVHIP_EINVAL, "ovhip_itx_launch_chroma_lmcs: null buffer", hipSuccess);
    if ((dst->stride_y & 7) || ((uintptr_t)dst->y & 15))
        return ov_fail(ctx, OVHIP_EINVAL, "ovhip_itx_launch_chroma_lmcs: luma plane must be 16-byte aligned with stride % 8 == 0", hipSuccess);
    return itx_launch(ctx, dst, d_cmds, n_large, n_small, d_coefs, d_lmcs_scales, d_bwd_lut);
}

/* library-internal (the picture job): any of the three launches above with the tiny tail of the small class named
 * (ovhip_rec_tb_cmds_split_tiny_); res / d_bwd_lut NULL when not used */
extern "C" int ovhip_itx_launch_ex_(ovhip_ctx *ctx, const ovhip_pic *dst, const ovhip_pic *res, const ovhip_tb_cmd *d_cmds,
                                    uint32_t n_large, uint32_t n_small, const uint32_t tiny3[4], const int16_t *d_coefs,
                                    const int16_t *d_lmcs_scales, const uint16_t *d_bwd_lut)
{
    if (!ctx || !dst) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    if (!n_large && !n_small && !d_bwd_lut) return OVHIP_OK;
    if ((n_large || n_small) && (!d_cmds || !d_coefs)) return ov_fail(ctx, OVHIP_EINVAL, "ovhip_itx_launch_ex_: null buffer", hipSuccess);
    if (d_bwd_lut && ((dst->stride_y & 7) || ((uintptr_t)dst->y & 15)))
        return ov_fail(ctx, OVHIP_EINVAL, "ovhip_itx_launch_ex_: luma plane must be 16-byte aligned with stride % 8 == 0", hipSuccess);
    return itx_launch(ctx, dst, d_cmds, n_large, n_small, d_coefs, d_lmcs_scales, d_bwd_lut, res, tiny3);
}

extern "C" int ovhip_itx_launch(ovhip_ctx *ctx, const ovhip_pic *dst, const ovhip_tb_cmd *d_cmds,
                                uint32_t n_cmds, const int16_t *d_coefs, const int16_t *d_lmcs_scales)
{
    return ovhip_itx_launch_classes(ctx, dst, d_cmds, n_cmds, 0, d_coefs, d_lmcs_scales);
}
