// Output path: conformance-window crop + pack into the byte layout examples/dectest.c:372-409 writes, and per-row
// MD5 digests on the device (see include/ovvc_hip.h, "Output path").
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "ovvc_hip.h"
#include "ovvc_common.hip.h"

namespace {

// The cropped planes as dectest.c:383-397 derives them: luma offsets are twice the window's (chroma-unit) offsets.
struct OutGeom {
    const uint16_t *src[3]; int stride[3], w[3], h[3];
    size_t first_sample[3], first_row[3];        // where the plane starts in the packed frame / in the row list
    size_t samples, rows;
};

static bool out_geom(int32_t w, int32_t h, const ovhip_window *win, const ovhip_pic *pic, OutGeom &g)
{
    const int l = win ? win->offset_lft : 0, r = win ? win->offset_rgt : 0, a = win ? win->offset_abv : 0, b = win ? win->offset_blw : 0;
    g.samples = 0; g.rows = 0;
    for (int c = 0; c < 3; ++c) {
        const int sh = c ? 0 : 1, pw = c ? w >> 1 : w, ph = c ? h >> 1 : h;
        g.w[c] = pw - ((l + r) << sh); g.h[c] = ph - ((a + b) << sh);
        if (g.w[c] <= 0 || g.h[c] <= 0) return false;
        g.first_sample[c] = g.samples; g.first_row[c] = g.rows;
        g.samples += (size_t)g.w[c] * g.h[c]; g.rows += (size_t)g.h[c];
        if (pic) {
            const uint16_t *p = c == 0 ? pic->y : (c == 1 ? pic->cb : pic->cr);
            g.stride[c] = c ? pic->stride_c : pic->stride_y;
            g.src[c] = p + (size_t)(a << sh) * g.stride[c] + (l << sh);
        }
    }
    return true;
}

__global__ __launch_bounds__(256) void k_output_pack(OutGeom g, uint16_t *__restrict__ out)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < g.samples; i += (size_t)gridDim.x * 256) {
        const int c = i >= g.first_sample[2] ? 2 : (i >= g.first_sample[1] ? 1 : 0);
        const size_t k = i - g.first_sample[c];
        const int y = (int)(k / (size_t)g.w[c]), x = (int)(k - (size_t)y * g.w[c]);
        out[i] = g.src[c][(size_t)y * g.stride[c] + x];
    }
}

__device__ __forceinline__ uint32_t rotl(uint32_t v, int s) { return (v << s) | (v >> (32 - s)); }

// one MD5 block (RFC 1321, the four rounds written out so that the message index and the shift are immediates)
__device__ __forceinline__ void md5_block(uint32_t h[4], const uint32_t m[16])
{
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3];
#define STEP(f, g, k, s) { const uint32_t t_ = (f) + a + (k) + m[g]; a = d; d = c; c = b; b += rotl(t_, s); }
#define F1 ((b & c) | (~b & d))
#define F2 ((d & b) | (~d & c))
#define F3 (b ^ c ^ d)
#define F4 (c ^ (b | ~d))
    STEP(F1, 0, 0xd76aa478u, 7)  STEP(F1, 1, 0xe8c7b756u, 12) STEP(F1, 2, 0x242070dbu, 17) STEP(F1, 3, 0xc1bdceeeu, 22)
    STEP(F1, 4, 0xf57c0fafu, 7)  STEP(F1, 5, 0x4787c62au, 12) STEP(F1, 6, 0xa8304613u, 17) STEP(F1, 7, 0xfd469501u, 22)
    STEP(F1, 8, 0x698098d8u, 7)  STEP(F1, 9, 0x8b44f7afu, 12) STEP(F1, 10, 0xffff5bb1u, 17) STEP(F1, 11, 0x895cd7beu, 22)
    STEP(F1, 12, 0x6b901122u, 7) STEP(F1, 13, 0xfd987193u, 12) STEP(F1, 14, 0xa679438eu, 17) STEP(F1, 15, 0x49b40821u, 22)
    STEP(F2, 1, 0xf61e2562u, 5)  STEP(F2, 6, 0xc040b340u, 9)  STEP(F2, 11, 0x265e5a51u, 14) STEP(F2, 0, 0xe9b6c7aau, 20)
    STEP(F2, 5, 0xd62f105du, 5)  STEP(F2, 10, 0x02441453u, 9) STEP(F2, 15, 0xd8a1e681u, 14) STEP(F2, 4, 0xe7d3fbc8u, 20)
    STEP(F2, 9, 0x21e1cde6u, 5)  STEP(F2, 14, 0xc33707d6u, 9) STEP(F2, 3, 0xf4d50d87u, 14) STEP(F2, 8, 0x455a14edu, 20)
    STEP(F2, 13, 0xa9e3e905u, 5) STEP(F2, 2, 0xfcefa3f8u, 9)  STEP(F2, 7, 0x676f02d9u, 14) STEP(F2, 12, 0x8d2a4c8au, 20)
    STEP(F3, 5, 0xfffa3942u, 4)  STEP(F3, 8, 0x8771f681u, 11) STEP(F3, 11, 0x6d9d6122u, 16) STEP(F3, 14, 0xfde5380cu, 23)
    STEP(F3, 1, 0xa4beea44u, 4)  STEP(F3, 4, 0x4bdecfa9u, 11) STEP(F3, 7, 0xf6bb4b60u, 16) STEP(F3, 10, 0xbebfbc70u, 23)
    STEP(F3, 13, 0x289b7ec6u, 4) STEP(F3, 0, 0xeaa127fau, 11) STEP(F3, 3, 0xd4ef3085u, 16) STEP(F3, 6, 0x04881d05u, 23)
    STEP(F3, 9, 0xd9d4d039u, 4)  STEP(F3, 12, 0xe6db99e5u, 11) STEP(F3, 15, 0x1fa27cf8u, 16) STEP(F3, 2, 0xc4ac5665u, 23)
    STEP(F4, 0, 0xf4292244u, 6)  STEP(F4, 7, 0x432aff97u, 10) STEP(F4, 14, 0xab9423a7u, 15) STEP(F4, 5, 0xfc93a039u, 21)
    STEP(F4, 12, 0x655b59c3u, 6) STEP(F4, 3, 0x8f0ccc92u, 10) STEP(F4, 10, 0xffeff47du, 15) STEP(F4, 1, 0x85845dd1u, 21)
    STEP(F4, 8, 0x6fa87e4fu, 6)  STEP(F4, 15, 0xfe2ce6e0u, 10) STEP(F4, 6, 0xa3014314u, 15) STEP(F4, 13, 0x4e0811a1u, 21)
    STEP(F4, 4, 0xf7537e82u, 6)  STEP(F4, 11, 0xbd3af235u, 10) STEP(F4, 2, 0x2ad7d2bbu, 15) STEP(F4, 9, 0xeb86d391u, 21)
#undef STEP
#undef F1
#undef F2
#undef F3
#undef F4
    h[0] += a; h[1] += b; h[2] += c; h[3] += d;
}

// MD5 of n samples (2 bytes each, little endian) starting at src
__device__ __forceinline__ void md5_samples(const uint16_t *src, int n, uint32_t h[4])
{
    h[0] = 0x67452301u; h[1] = 0xefcdab89u; h[2] = 0x98badcfeu; h[3] = 0x10325476u;
    uint32_t m[16];
    int s = 0;
    if (((uintptr_t)src & 3) == 0) {
        const uint32_t *s32 = reinterpret_cast<const uint32_t *>(src);
        for (; s + 32 <= n; s += 32) {
#pragma unroll
            for (int i = 0; i < 16; ++i) m[i] = s32[(s >> 1) + i];
            md5_block(h, m);
        }
    } else {
        for (; s + 32 <= n; s += 32) {
#pragma unroll
            for (int i = 0; i < 16; ++i) m[i] = (uint32_t)src[s + 2 * i] | ((uint32_t)src[s + 2 * i + 1] << 16);
            md5_block(h, m);
        }
    }
    // tail: rem samples (< 32), the 0x80 byte, zeros, the message length in bits
    const int rem = n - s;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int s0 = 2 * i, s1 = 2 * i + 1;
        const uint32_t lo = s0 < rem ? src[s + s0] : (s0 == rem ? 0x80u : 0u);
        const uint32_t hi = s1 < rem ? src[s + s1] : (s1 == rem ? 0x80u : 0u);
        m[i] = lo | (hi << 16);
    }
    const uint64_t bits = (uint64_t)n * 16;
    if (2 * rem + 1 > 56) {
        md5_block(h, m);
#pragma unroll
        for (int i = 0; i < 16; ++i) m[i] = 0;
    }
    m[14] = (uint32_t)bits; m[15] = (uint32_t)(bits >> 32);
    md5_block(h, m);
}

// MD5 of nd 16-byte digests held as words (LDS)
__device__ __forceinline__ void md5_digests(const uint32_t *w, int nd, uint32_t h[4])
{
    h[0] = 0x67452301u; h[1] = 0xefcdab89u; h[2] = 0x98badcfeu; h[3] = 0x10325476u;
    uint32_t m[16];
    int d = 0;
    for (; d + 4 <= nd; d += 4) {
#pragma unroll
        for (int i = 0; i < 16; ++i) m[i] = w[4 * d + i];
        md5_block(h, m);
    }
    const int rem = nd - d;                       // 0..3 digests = 0..48 bytes: the padding always fits this block
#pragma unroll
    for (int i = 0; i < 16; ++i) m[i] = i < 4 * rem ? w[4 * d + i] : (i == 4 * rem ? 0x80u : 0u);
    const uint64_t bits = (uint64_t)nd * 128;
    m[14] = (uint32_t)bits; m[15] = (uint32_t)(bits >> 32);
    md5_block(h, m);
}

// ---- the tree fingerprint's own hash (include/ovvc_hip.h, "Digest"): 128 bits, four FNV-1a lanes over 32-bit words with murmur3's
// finaliser -- a 512-byte leaf is ~300 integer operations where MD5 took ~2900 (round 4: the MD5 tree was 6.5 M vector instructions
// per 4K picture, more than k_itx_all, and cost the stream 15 % beside `output none`).  Every step is a bijection of its lane for a
// given word, so any change of one word changes the digest; the FILE's MD5 is a different thing (OVHIP_STREAM_FILE_MD5).
__device__ __forceinline__ uint32_t mix_fin(uint32_t h)
{
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
#define MIX_STEP(hj, w) (hj) = ((hj) ^ (w)) * 0x01000193u
__device__ __forceinline__ void mix_begin(uint32_t h[4], uint32_t tag)
{
    h[0] = 0x67452301u ^ tag; h[1] = 0xefcdab89u; h[2] = 0x98badcfeu; h[3] = 0x10325476u;
}
__device__ __forceinline__ void mix_end(uint32_t h[4])
{
    h[0] = mix_fin(h[0]); h[1] = mix_fin(h[1]); h[2] = mix_fin(h[2]); h[3] = mix_fin(h[3]);
    h[0] += h[1]; h[2] += h[3]; h[0] += h[2]; h[1] += h[0]; h[2] += h[0]; h[3] += h[0];
}
// n samples (little endian, two per word; an odd last sample alone in its word), tag = n
__device__ __forceinline__ void mix_samples(const uint16_t *src, int n, uint32_t h[4])
{
    mix_begin(h, (uint32_t)n);
    int s = 0;
    if (((uintptr_t)src & 15) == 0) {
        const uint4 *s128 = reinterpret_cast<const uint4 *>(src);
        for (; s + 8 <= n; s += 8) {
            const uint4 v = s128[s >> 3];
            MIX_STEP(h[0], v.x); MIX_STEP(h[1], v.y); MIX_STEP(h[2], v.z); MIX_STEP(h[3], v.w);
        }
    } else {
        for (; s + 8 <= n; s += 8) {
#pragma unroll
            for (int j = 0; j < 4; ++j) MIX_STEP(h[j], (uint32_t)src[s + 2 * j] | ((uint32_t)src[s + 2 * j + 1] << 16));
        }
    }
    for (int j = 0; s < n; s += 2, ++j) MIX_STEP(h[j], (uint32_t)src[s] | (s + 1 < n ? (uint32_t)src[s + 1] << 16 : 0u));      // (s is a multiple of 8 here: word index & 3 == j)
    mix_end(h);
}
// nd 16-byte digests held as words, tag = nd
__device__ __forceinline__ void mix_digests(const uint32_t *w, int nd, uint32_t h[4])
{
    mix_begin(h, (uint32_t)nd);
    for (int d = 0; d < nd; ++d) { MIX_STEP(h[0], w[4 * d]); MIX_STEP(h[1], w[4 * d + 1]); MIX_STEP(h[2], w[4 * d + 2]); MIX_STEP(h[3], w[4 * d + 3]); }
    mix_end(h);
}

// One lane per cropped row: the row's samples are its message.  A wave reads 64 rows side by side;
// every lane walks its own row, which stays in the L1 of the compute unit between its loads.
__global__ __launch_bounds__(64) void k_output_row_md5(OutGeom g, uint8_t *__restrict__ digests)
{
    const size_t row = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (row >= g.rows) return;
    const int c = row >= g.first_row[2] ? 2 : (row >= g.first_row[1] ? 1 : 0);
    uint32_t h[4];
    md5_samples(g.src[c] + (row - g.first_row[c]) * (size_t)g.stride[c], g.w[c], h);
    uint32_t *o = reinterpret_cast<uint32_t *>(digests + row * 16);
    o[0] = h[0]; o[1] = h[1]; o[2] = h[2]; o[3] = h[3];
}

// The picture's fingerprint as a three-level hash tree (include/ovvc_hip.h, "Digest"; until round 4 every level was MD5, now only the
// host's last one is): MD5 is a serial chain, a whole row per lane is
// 120 chained blocks at 4K (the one-lane-per-row kernel above takes ~200 us, on every picture's stream).  Leaves = the 512-byte
// pieces of every cropped row (8 + 1 blocks), then one digest per row over its pieces' digests (4 blocks at 4K), then one per band of
// 8 rows (3 blocks): a 128-thread workgroup per band does all three levels through LDS -- 16 chained blocks and 540 workgroups at
// 4K (the first shape, 1024-byte leaves and bands of 32 rows: 28 chained blocks, 136 workgroups on 256 compute units).  The host
// hashes the band digests (8.6 KB at 4K).
#define DG_SEG   256          /* samples per leaf (512 bytes)  */
#define DG_BAND  8            /* rows per band                  */
#define DG_MAXSEG 64          /* leaves per row: rows up to 16384 samples */
#define DG_NT    128
__global__ __launch_bounds__(DG_NT) void k_output_tree_md5(OutGeom g, uint32_t nb0, uint32_t nb1, uint8_t *__restrict__ digests)
{
    __shared__ uint32_t s_d0[DG_BAND * DG_MAXSEG * 4];
    __shared__ uint32_t s_d1[DG_BAND * 4];
    const uint32_t band = blockIdx.x;                      // bands of Y, then Cb, then Cr
    const int c = band >= nb0 + nb1 ? 2 : (band >= nb0 ? 1 : 0);
    const int r0 = (int)(band - (c == 2 ? nb0 + nb1 : (c == 1 ? nb0 : 0))) * DG_BAND;
    const int nrows = min(DG_BAND, g.h[c] - r0), w = g.w[c], nseg = (w + DG_SEG - 1) / DG_SEG;
    const int tid = threadIdx.x;
    for (int i = tid; i < nrows * nseg; i += DG_NT) {
        const int r = i / nseg, k = i - r * nseg;
        uint32_t h[4];
        mix_samples(g.src[c] + (size_t)(r0 + r) * g.stride[c] + k * DG_SEG, min(DG_SEG, w - k * DG_SEG), h);
        uint32_t *o = s_d0 + (r * DG_MAXSEG + k) * 4;
        o[0] = h[0]; o[1] = h[1]; o[2] = h[2]; o[3] = h[3];
    }
    __syncthreads();
    if (tid < nrows) {
        uint32_t h[4];
        mix_digests(s_d0 + tid * DG_MAXSEG * 4, nseg, h);
        s_d1[4 * tid] = h[0]; s_d1[4 * tid + 1] = h[1]; s_d1[4 * tid + 2] = h[2]; s_d1[4 * tid + 3] = h[3];
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t h[4];
        mix_digests(s_d1, nrows, h);
        uint32_t *o = reinterpret_cast<uint32_t *>(digests + (size_t)band * 16);
        o[0] = h[0]; o[1] = h[1]; o[2] = h[2]; o[3] = h[3];
    }
}

} // namespace

extern "C" size_t ovhip_output_bytes(int32_t w, int32_t h, const ovhip_window *win)
{
    OutGeom g;
    if (w <= 0 || h <= 0 || (w & 1) || (h & 1) || !out_geom(w, h, win, nullptr, g)) return 0;
    return g.samples * sizeof(uint16_t);
}

extern "C" size_t ovhip_output_rows(int32_t w, int32_t h, const ovhip_window *win)
{
    OutGeom g;
    if (w <= 0 || h <= 0 || (w & 1) || (h & 1) || !out_geom(w, h, win, nullptr, g)) return 0;
    return g.rows;
}

extern "C" int ovhip_output_pack_launch(ovhip_ctx *ctx, const ovhip_pic *pic, const ovhip_window *win, uint16_t *d_out)
{
    if (!ctx || !pic) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    OutGeom g;
    if (!d_out || pic->w <= 0 || pic->h <= 0 || (pic->w & 1) || (pic->h & 1) || !out_geom(pic->w, pic->h, win, pic, g))
        return ov_fail(ctx, OVHIP_EINVAL, "ovhip_output_pack_launch: bad picture / window", hipSuccess);
    const size_t wgs = (g.samples + 255) / 256;
    hipLaunchKernelGGL(k_output_pack, dim3((unsigned)(wgs < 65536 ? wgs : 65536)), dim3(256), 0, ctx->stream, g, d_out);
    OV_LAUNCH_CHECK(ctx, "k_output_pack");
    return OVHIP_OK;
}

extern "C" int ovhip_output_row_md5_launch(ovhip_ctx *ctx, const ovhip_pic *pic, const ovhip_window *win, uint8_t *d_digests)
{
    if (!ctx || !pic) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    OutGeom g;
    if (!d_digests || ((uintptr_t)d_digests & 3) || pic->w <= 0 || pic->h <= 0 || (pic->w & 1) || (pic->h & 1) || !out_geom(pic->w, pic->h, win, pic, g))
        return ov_fail(ctx, OVHIP_EINVAL, "ovhip_output_row_md5_launch: bad picture / window", hipSuccess);
    hipLaunchKernelGGL(k_output_row_md5, dim3((unsigned)((g.rows + 63) / 64)), dim3(64), 0, ctx->stream, g, d_digests);
    OV_LAUNCH_CHECK(ctx, "k_output_row_md5");
    return OVHIP_OK;
}

extern "C" int ovhip_pic_output(ovhip_ctx *ctx, const ovhip_pic *pic, const ovhip_window *win, void *host_dst)
{
    if (!ctx || !pic || !host_dst) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    const size_t bytes = ovhip_output_bytes(pic->w, pic->h, win);
    if (!bytes) return ov_fail(ctx, OVHIP_EINVAL, "ovhip_pic_output: bad picture / window", hipSuccess);
    int r = ov_scratch(ctx, bytes, 0);
    if (r) return r;
    uint16_t *d = (uint16_t *)ctx->scratch_d;
    r = ovhip_output_pack_launch(ctx, pic, win, d);
    if (!r && hipMemcpyAsync(host_dst, d, bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) r = ov_fail(ctx, OVHIP_ENODEV, "ovhip_pic_output: D2H", hipGetLastError());
    if (!r && ov_sync_stream(ctx) != hipSuccess) r = ov_fail(ctx, OVHIP_ELAUNCH, "ovhip_pic_output", hipGetLastError());
    return r;
}

// Bands of the digest tree: 16 bytes each in d_digests (Y bands, Cb bands, Cr bands).
extern "C" size_t ovhip_output_bands(int32_t w, int32_t h, const ovhip_window *win)
{
    OutGeom g;
    if (w <= 0 || h <= 0 || (w & 1) || (h & 1) || !out_geom(w, h, win, nullptr, g) || g.w[0] > DG_SEG * DG_MAXSEG) return 0;
    return (size_t)(g.h[0] + DG_BAND - 1) / DG_BAND + 2 * (size_t)((g.h[1] + DG_BAND - 1) / DG_BAND);
}

extern "C" int ovhip_output_tree_md5_launch(ovhip_ctx *ctx, const ovhip_pic *pic, const ovhip_window *win, uint8_t *d_digests)
{
    if (!ctx || !pic) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    OutGeom g;
    if (!d_digests || ((uintptr_t)d_digests & 3) || pic->w <= 0 || pic->h <= 0 || (pic->w & 1) || (pic->h & 1) || !out_geom(pic->w, pic->h, win, pic, g)
        || g.w[0] > DG_SEG * DG_MAXSEG)
        return ov_fail(ctx, OVHIP_EINVAL, "ovhip_output_tree_md5_launch: bad picture / window", hipSuccess);
    const uint32_t nb0 = (uint32_t)((g.h[0] + DG_BAND - 1) / DG_BAND), nb1 = (uint32_t)((g.h[1] + DG_BAND - 1) / DG_BAND);
    hipLaunchKernelGGL(k_output_tree_md5, dim3(nb0 + 2 * nb1), dim3(DG_NT), 0, ctx->stream, g, nb0, nb1, d_digests);
    OV_LAUNCH_CHECK(ctx, "k_output_tree_md5");
    return OVHIP_OK;
}

extern "C" int ovhip_pic_digest(ovhip_ctx *ctx, const ovhip_pic *pic, const ovhip_window *win, uint8_t out[16])
{
    if (!ctx || !pic || !out) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    const size_t nb = ovhip_output_bands(pic->w, pic->h, win);
    if (!nb) return ov_fail(ctx, OVHIP_EINVAL, "ovhip_pic_digest: bad picture / window", hipSuccess);
    int r = ov_scratch(ctx, nb * 16, nb * 16);
    if (r) return r;
    // the band digests are stored straight into page-locked host memory by the kernel (write-only, 16 bytes per band): no DMA hop
    // between the kernel and the host's wait
    uint8_t *hbuf = (uint8_t *)ctx->scratch_h;
    r = ovhip_output_tree_md5_launch(ctx, pic, win, hbuf);
    if (!r && ov_sync_stream(ctx) != hipSuccess) r = ov_fail(ctx, OVHIP_ELAUNCH, "ovhip_pic_digest", hipGetLastError());
    if (!r) { ovhip_md5_state st; ovhip_md5_init(&st); ovhip_md5_update(&st, hbuf, nb * 16); ovhip_md5_final(&st, out); }
    return r;
}
