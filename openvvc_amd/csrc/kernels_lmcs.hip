// kernels_lmcs.hip -- K11 on gfx950: the two per-sample / per-region LMCS steps that need the
// reconstructed picture (the forward map is fused into the MC kernels, the tables are built on the
// host by ovvc_lmcs.c).
//
//   k_lmcs_scale   : rcn_lmcs_compute_chroma_scale (libovvc/rcn_lmcs.c:204-350), called by the
//                    reference once per 64-aligned CU (vcl_coding_unit.c:724-730): average of the
//                    <= 64 + 64 reshaped-domain luma samples above / left of the region (padded with
//                    the last available sample to 16 units per side), window lookup, division.
//                    One wavefront per region: lane = neighbour sample, wave reduction.
//   k_lmcs_inverse : lmcs_reshape_backward over the whole luma plane (slicedec.c:746-750): one 16-byte
//                    vector of 8 samples per lane, LUT (2 KB) staged in LDS.  Pure HBM stream: 2 bytes
//                    read + 2 bytes written per luma sample.
#include "ovvc_common.hip.h"
#include "flow_state.hip.h"

namespace {

struct LmcsWnd { uint16_t bnd[17]; int min_idx, max_idx, crs_offset; };
#ifndef LMCS_ROWS
#define LMCS_ROWS 4
#endif

// Workgroups [0, n): one chroma-scale region each.  Workgroups beyond: rider -- the state words of the flow launch of the ordered
// pass (k_intra_flow_prepare's work, four tasks per workgroup), when the picture has one: a launch of its own was 9 us of the
// picture's launch chain.
__global__ __launch_bounds__(64) void k_lmcs_scale(ovhip_pic pic, const ovhip_lmcs_region *__restrict__ regs, uint32_t n,
                                                   LmcsWnd wnd, int16_t *__restrict__ scales, const ovhip_itask *__restrict__ tasks,
                                                   uint32_t n_tasks, FlowState fs, unsigned epoch)
{
    const uint32_t bid = blockIdx.x;
    if (bid >= n) {
        const uint32_t ti = (bid - n) * 4 + (threadIdx.x >> 4);
        if (ti < n_tasks) flow_prepare_task(tasks[ti], fs, epoch, threadIdx.x & 15);
        return;
    }
    const ovhip_lmcs_region g = regs[bid];
    if (g.ordered) return;                                    // luma around it comes from ordered tasks: k_intra_level derives it
    const int lane = threadIdx.x;
    // the reference sums sample k of every unit into luma_sum[k & 3]; only the total is used
    const uint16_t *src = pic.y + (size_t)g.y * pic.stride_y + g.x;
    // both loads unconditional (a side that does not exist reads the region's own first sample and is dropped): inside the
    // branches they were two round trips, one after the other
    const int na = 4 * g.n_abv, nl = 4 * g.n_lft;            // samples actually read; the rest repeats the last one
    const long oa = g.n_abv ? (long)min(lane, na - 1) - pic.stride_y : 0;
    const long ol = g.n_lft ? (long)min(lane, nl - 1) * pic.stride_y - 1 : 0;
    const int va = src[oa], vl = src[ol];
    int sum = (g.n_abv ? va : 0) + (g.n_lft ? vl : 0);
#pragma unroll
    for (int m = 32; m; m >>= 1) sum += __shfl_xor(sum, m);
    if (lane == 0) {
        const int nb_units = (g.n_abv ? 16 : 0) + (g.n_lft ? 16 : 0);
        int log2_nb = 0;
        for (int v = nb_units; v; v >>= 1) ++log2_nb;         // 16 -> 5, 32 -> 6, as the reference counts
        const int avg = log2_nb ? (sum + (1 << log2_nb)) >> (log2_nb + 1) : 512;
        int idx = wnd.min_idx;                                // get_bwd_idx (rcn_lmcs.c:83-93)
        for (; idx < wnd.max_idx; ++idx)
            if (avg < wnd.bnd[idx + 1]) break;
        idx = min(idx, 15);
        const int wnd_sz = (int)wnd.bnd[idx + 1] - (int)wnd.bnd[idx];
        scales[bid] = (int16_t)(wnd_sz == 0 ? 1 << 11 : (1 << (OV_BD - 4 + 11)) / (wnd_sz + wnd.crs_offset));
    }
}

typedef uint32_t lm_u2 __attribute__((ext_vector_type(2), aligned(4)));

// k_flow_untag's body (kernels_intra.hip) for the chroma blocks of the ordered tasks: rides in the inverse-mapping launch of a picture
// whose ordered pass ran as flow launches (one kernel boundary less between the ordered pass and the deblocking filter).
// One wave per task, four tasks per workgroup; a lane's loads of BOTH planes go out before anything is stored: with 16 lanes per task
// and a load - and - store round trip per 64 samples and plane the rider was 8.6 of the launch's 16.7 us (the mapping alone: 8.0).
#define UNTAG_TASKS_PER_WG 4
__device__ __forceinline__ void untag_chroma_tasks(const ovhip_pic &pic, const ovhip_itask *__restrict__ tasks, uint32_t n, uint32_t wg)
{
    const uint32_t ti = wg * UNTAG_TASKS_PER_WG + (threadIdx.x >> 6);
    if (ti >= n) return;
    const ovhip_itask t = tasks[ti];
    if (t.kind == OVHIP_IT_REGION || t.kind == OVHIP_IT_LUMA) return;          // (the mapping itself drops the bit of every luma sample)
    const int lane = threadIdx.x & 63;
    const int l2w = t.log2_w, w = 1 << l2w, npx = w << t.log2_h, stride = pic.stride_c;
    const bool res_only = t.kind == OVHIP_IT_RES_C;
    const bool on0 = !res_only || (t.flags & OVHIP_IF_RES_CB), on1 = !res_only || (t.flags & OVHIP_IF_RES_CR);
    uint16_t *const d0 = pic.cb + t.y * stride + t.x, *const d1 = pic.cr + t.y * stride + t.x;
    if (w >= 4) {
        for (int base = 0; base < npx; base += 1024) {
            lm_u2 v0[4], v1[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int p = base + 4 * lane + 256 * k;
                const int o = (p >> l2w) * stride + (p & (w - 1));
                if (p < npx && on0) v0[k] = *reinterpret_cast<const lm_u2 *>(d0 + o);
                if (p < npx && on1) v1[k] = *reinterpret_cast<const lm_u2 *>(d1 + o);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int p = base + 4 * lane + 256 * k;
                const int o = (p >> l2w) * stride + (p & (w - 1));
                if (p < npx && on0) { lm_u2 v = v0[k]; v[0] &= 0x03ff03ffu; v[1] &= 0x03ff03ffu; *reinterpret_cast<lm_u2 *>(d0 + o) = v; }
                if (p < npx && on1) { lm_u2 v = v1[k]; v[0] &= 0x03ff03ffu; v[1] &= 0x03ff03ffu; *reinterpret_cast<lm_u2 *>(d1 + o) = v; }
            }
        }
    } else if (w == 2) {
        for (int p = 2 * lane; p < npx; p += 128) {
            const int o = (p >> l2w) * stride + (p & (w - 1));
            uint32_t a = 0, b = 0;
            if (on0) a = *reinterpret_cast<const uint32_t *>(d0 + o);
            if (on1) b = *reinterpret_cast<const uint32_t *>(d1 + o);
            if (on0) *reinterpret_cast<uint32_t *>(d0 + o) = a & 0x03ff03ffu;
            if (on1) *reinterpret_cast<uint32_t *>(d1 + o) = b & 0x03ff03ffu;
        }
    } else {
        for (int p = lane; p < npx; p += 64) {
            const int o = (p >> l2w) * stride + (p & (w - 1));
            if (on0) d0[o] = d0[o] & 0x3ff;
            if (on1) d1[o] = d1[o] & 0x3ff;
        }
    }
}

// grid.y = rows_y row groups of the mapping + the workgroups of the chroma un-tag (grid.x of them per y)
__global__ __launch_bounds__(256) void k_lmcs_inverse(ovhip_pic pic, const uint16_t *__restrict__ lut, uint32_t rows_y,
                                                      const ovhip_itask *__restrict__ tasks, uint32_t n_tasks)
{
    // the un-tag workgroups come FIRST in the grid: they are short read-modify-writes of scattered chroma blocks (latency), and
    // started last they were the tail of the launch behind the streaming part
    const uint32_t untag_y = gridDim.y - rows_y;
    if (blockIdx.y < untag_y) { untag_chroma_tasks(pic, tasks, n_tasks, blockIdx.y * gridDim.x + blockIdx.x); return; }
    const uint32_t by = blockIdx.y - untag_y;

    __shared__ uint16_t s_lut[1024];
    const int nvx = pic.w >> 3;                               // full 8-sample vectors per row
    const int tail = pic.w & 7;
    // a workgroup takes LMCS_ROWS rows at a time (the LUT staged once for 16 KB of samples instead of once per row segment).  The
    // lane's vectors of its FIRST row group are requested before the table is staged: the two round trips (table, samples) overlap
    // instead of following each other behind the barrier (20 -> see DESIGN 4 us per 4K picture, where this launch stands alone behind
    // the ordered pass)
    const int y00 = (int)by * LMCS_ROWS, v00 = blockIdx.x * 256 + threadIdx.x;
    uint4 q0[LMCS_ROWS];
    const bool first = y00 < pic.h && v00 < nvx;
    if (first) {
#pragma unroll
        for (int r = 0; r < LMCS_ROWS; ++r) q0[r] = *reinterpret_cast<const uint4 *>(pic.y + (size_t)min(y00 + r, pic.h - 1) * pic.stride_y + 8 * v00);
    }
    for (int i = threadIdx.x; i < 512; i += 256) reinterpret_cast<uint32_t *>(s_lut)[i] = reinterpret_cast<const uint32_t *>(lut)[i];
    __syncthreads();
    for (int y0 = y00; y0 < pic.h; y0 += rows_y * LMCS_ROWS) {
        for (int v = v00; v < nvx; v += gridDim.x * 256) {
            uint4 q[LMCS_ROWS];
            if (y0 == y00 && v == v00) {
#pragma unroll
                for (int r = 0; r < LMCS_ROWS; ++r) q[r] = q0[r];
            } else {
#pragma unroll
                for (int r = 0; r < LMCS_ROWS; ++r) q[r] = *reinterpret_cast<const uint4 *>(pic.y + (size_t)min(y0 + r, pic.h - 1) * pic.stride_y + 8 * v);
            }
#pragma unroll
            for (int r = 0; r < LMCS_ROWS; ++r) {
                if (y0 + r >= pic.h) break;
                uint32_t *d = reinterpret_cast<uint32_t *>(&q[r]);
#pragma unroll
                for (int k = 0; k < 4; ++k) d[k] = s_lut[d[k] & 1023] | ((uint32_t)s_lut[(d[k] >> 16) & 1023] << 16);
                *reinterpret_cast<uint4 *>(pic.y + (size_t)(y0 + r) * pic.stride_y + 8 * v) = q[r];
            }
        }
        if (tail && blockIdx.x == 0 && threadIdx.x < tail)
            for (int r = 0; r < LMCS_ROWS && y0 + r < pic.h; ++r) {
                uint16_t *row = pic.y + (size_t)(y0 + r) * pic.stride_y;
                row[8 * nvx + threadIdx.x] = s_lut[row[8 * nvx + threadIdx.x] & 1023];
            }
    }
}

} // namespace

static int lmcs_scale_launch(ovhip_ctx *ctx, const ovhip_pic *pic, const ovhip_lmcs_region *d_regions, uint32_t n_regions, const ovhip_lmcs_luts *luts,
                             int16_t *d_scales, const ovhip_itask *d_tasks, uint32_t n_tasks, uint32_t *d_state, uint32_t epoch)
{
    if (!ctx || !pic || !luts) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    if (!n_regions && !n_tasks) return OVHIP_OK;
    if (n_regions && (!d_regions || !d_scales)) return ov_fail(ctx, OVHIP_EINVAL, "ovhip_lmcs_scale_launch: null buffer", hipSuccess);
    if (n_tasks && (!d_tasks || !d_state || !epoch)) return ov_fail(ctx, OVHIP_EINVAL, "ovhip_lmcs_scale_prepare_launch: bad arguments", hipSuccess);
    LmcsWnd w;
    for (int i = 0; i < 17; ++i) w.bnd[i] = luts->wnd_bnd[i];
    w.min_idx = luts->min_idx; w.max_idx = luts->max_idx; w.crs_offset = luts->crs_offset;
    FlowState fs;
    memset(&fs, 0, sizeof(fs));
    if (n_tasks) fs = flow_state_of(d_state, pic->w, pic->h);
    hipLaunchKernelGGL(k_lmcs_scale, dim3(n_regions + (n_tasks + 3) / 4), dim3(64), 0, ctx->stream, *pic, d_regions, n_regions, w, d_scales, d_tasks, n_tasks, fs, epoch);
    OV_LAUNCH_CHECK(ctx, n_tasks ? "k_lmcs_scale (+ flow state words)" : "k_lmcs_scale");
    return OVHIP_OK;
}

extern "C" int ovhip_lmcs_scale_launch(ovhip_ctx *ctx, const ovhip_pic *pic, const ovhip_lmcs_region *d_regions,
                                       uint32_t n_regions, const ovhip_lmcs_luts *luts, int16_t *d_scales)
{
    return lmcs_scale_launch(ctx, pic, d_regions, n_regions, luts, d_scales, nullptr, 0, nullptr, 0);
}

// ovhip_lmcs_scale_launch + the state words of the picture's flow launch (what ovhip_intra_flow_launch does first when called with
// prepare != 0) in ONE launch: d_tasks[n_tasks] = the level-sorted ordered tasks, d_state / epoch as for ovhip_intra_flow_launch,
// whose launches of this picture are then called with prepare = 0.
extern "C" int ovhip_lmcs_scale_prepare_launch(ovhip_ctx *ctx, const ovhip_pic *pic, const ovhip_lmcs_region *d_regions, uint32_t n_regions,
                                               const ovhip_lmcs_luts *luts, int16_t *d_scales, const ovhip_itask *d_tasks, uint32_t n_tasks,
                                               uint32_t *d_state, uint32_t epoch)
{
    return lmcs_scale_launch(ctx, pic, d_regions, n_regions, luts, d_scales, d_tasks, n_tasks, d_state, epoch);
}

static int lmcs_inverse_launch(ovhip_ctx *ctx, const ovhip_pic *pic, const uint16_t *d_bwd_lut, const ovhip_itask *d_tasks, uint32_t n_tasks, const char *who)
{
    if (!ctx || !pic || !d_bwd_lut) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    if ((pic->stride_y & 7) || ((uintptr_t)pic->y & 15))
        return ov_fail(ctx, OVHIP_EINVAL, "ovhip_lmcs_inverse_launch: luma plane must be 16-byte aligned with stride % 8 == 0", hipSuccess);
    const int nvx = pic->w >> 3;
    const int row_groups = (pic->h + LMCS_ROWS - 1) / LMCS_ROWS;
    const uint32_t gx = (nvx + 255) / 256 > 0 ? (nvx + 255) / 256 : 1, rows_y = row_groups < 4096 ? row_groups : 4096;
    const uint32_t untag_wgs = d_tasks ? (n_tasks + UNTAG_TASKS_PER_WG - 1) / UNTAG_TASKS_PER_WG : 0;
    hipLaunchKernelGGL(k_lmcs_inverse, dim3(gx, rows_y + (untag_wgs + gx - 1) / gx), dim3(256), 0, ctx->stream, *pic, d_bwd_lut, rows_y, d_tasks, d_tasks ? n_tasks : 0u);
    OV_LAUNCH_CHECK(ctx, who);
    return OVHIP_OK;
}

extern "C" int ovhip_lmcs_inverse_launch(ovhip_ctx *ctx, const ovhip_pic *pic, const uint16_t *d_bwd_lut)
{
    return lmcs_inverse_launch(ctx, pic, d_bwd_lut, nullptr, 0, "k_lmcs_inverse");
}

// The inverse mapping of a picture whose ordered pass ran as flow launches: the same launch also clears the hand-over bit
// (ovhip_intra_flow_untag_launch) in the chroma blocks of the ordered tasks d_tasks[n_tasks] (DEVICE); the mapping itself drops it
// from every luma sample.
extern "C" int ovhip_lmcs_inverse_untag_launch(ovhip_ctx *ctx, const ovhip_pic *pic, const uint16_t *d_bwd_lut, const ovhip_itask *d_tasks, uint32_t n_tasks)
{
    if (n_tasks && !d_tasks) return OVHIP_EINVAL;
    return lmcs_inverse_launch(ctx, pic, d_bwd_lut, n_tasks ? d_tasks : nullptr, n_tasks, "k_lmcs_inverse (+ chroma un-tag)");
}
