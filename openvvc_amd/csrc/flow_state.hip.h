// State words of the flow launch of the ordered pass (kernels_intra.hip) -- shared with the launch that prepares them as a rider
// (kernels_lmcs.hip).  Layout of the block ovhip_intra_flow_words() sizes: OVHIP_FLOW_SYNC_WORDS control words (word 0 = abort
// code), then one word per 4x4-luma unit for Y, Cb, Cr, then the chroma-scale regions' words.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ovvc_hip.h"

#define OVHIP_FLOW_SYNC_WORDS 16
struct FlowState { unsigned *y, *c[2], *reg; int w4; };

static inline FlowState flow_state_of(uint32_t *d_state, int w, int h)
{
    FlowState fs;
    const size_t nu = (size_t)((w + 3) / 4) * ((h + 3) / 4);
    fs.w4 = (w + 3) / 4;
    fs.y = d_state + OVHIP_FLOW_SYNC_WORDS; fs.c[0] = fs.y + nu; fs.c[1] = fs.c[0] + nu; fs.reg = fs.c[1] + nu;
    return fs;
}

// 2 * epoch = "an ordered task of this picture will write this unit" into the words of the units task t covers; 16 lanes per task
__device__ __forceinline__ void flow_prepare_task(const ovhip_itask &t, const FlowState &fs, unsigned epoch, int lane16)
{
    const unsigned mark = 2 * epoch;
    if (t.kind == OVHIP_IT_REGION) { if (lane16 == 0) fs.reg[t.c_scale] = mark; return; }
    const bool luma = t.kind == OVHIP_IT_LUMA;
    const int sh = luma ? 2 : 1, w = 1 << t.log2_w, h = 1 << t.log2_h;
    const int ux0 = t.x >> sh, uy0 = t.y >> sh, nx = max(1, w >> sh), ny = max(1, h >> sh);
    const int l2nx = 31 - __clz(nx);
    for (int i = lane16; i < nx * ny; i += 16) {
        const int u = (uy0 + (i >> l2nx)) * fs.w4 + ux0 + (i & (nx - 1));
        if (luma) fs.y[u] = mark;
        else {
            if (t.kind == OVHIP_IT_CHROMA || (t.flags & OVHIP_IF_RES_CB)) fs.c[0][u] = mark;
            if (t.kind == OVHIP_IT_CHROMA || (t.flags & OVHIP_IF_RES_CR)) fs.c[1][u] = mark;
        }
    }
}
