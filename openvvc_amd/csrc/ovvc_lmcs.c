/* ovvc_lmcs.c -- host side of K11: the LMCS look-up tables of one APS.
 *
 * Restates rcn_init_lmcs -> rcn_lmcs_compute_lut_luma -> init_lmcs_lut (libovvc/rcn_lmcs.c:93-179,
 * :303-311, :352-370) for 10-bit samples: 16 windows of 64 input codewords each; window i of the
 * mapped domain spans [wnd_bnd[i], wnd_bnd[i+1]) with wnd_bnd[i+1] - wnd_bnd[i] = 64 + cw_delta[i].
 * The tables are tiny (2 x 1024 uint16) and change once per picture at most, so they are built on
 * the host and uploaded; the per-sample work (forward map in the MC kernels, inverse map and the
 * chroma-scale derivation in kernels_lmcs.hip) is on the device.
 */
#include <string.h>
#include "ovvc_hip.h"

#define BD 10
#define LOG2_NB_WND 4
#define NB_WND 16
#define NB_SMP_WND (1 << (BD - LOG2_NB_WND))
#define LOG2_WND_RNG (BD - LOG2_NB_WND)
#define LMCS_PREC 11

static int clip_bd(int v) { return v < 0 ? 0 : v > 1023 ? 1023 : v; }

int
ovhip_lmcs_build(const ovhip_lmcs_data *data, ovhip_lmcs_luts *out)
{
    if (!data || !out || data->min_bin_idx >= NB_WND || data->delta_max_bin_idx >= NB_WND) return OVHIP_EINVAL;
    const int min_idx = data->min_bin_idx, max_idx_plus1 = NB_WND - data->delta_max_bin_idx;
    uint16_t fwd_step[NB_WND], bwd_step[NB_WND], wnd_bnd[NB_WND + 1];
    int16_t cw[NB_WND];

    /* lmcs_convert_data_to_info: deltas outside [min, max) are ignored */
    memset(cw, 0, sizeof(cw));
    for (int i = min_idx; i < max_idx_plus1; ++i) cw[i] = data->cw_delta[i];

    /* compute_windows_scale_steps */
    memset(wnd_bnd, 0, sizeof(wnd_bnd));
    memset(fwd_step, 0, sizeof(fwd_step));
    memset(bwd_step, 0, sizeof(bwd_step));
    for (int i = min_idx; i < max_idx_plus1; ++i) {
        int32_t wnd_sz = NB_SMP_WND + cw[i];
        if (wnd_sz) {
            fwd_step[i] = (uint16_t)(((wnd_sz << LMCS_PREC) + (1 << (LOG2_WND_RNG - 1))) >> LOG2_WND_RNG);
            bwd_step[i] = (uint16_t)((NB_SMP_WND << LMCS_PREC) / wnd_sz);
        }
        wnd_bnd[i + 1] = (uint16_t)(wnd_bnd[i] + wnd_sz);
    }
    for (int i = max_idx_plus1; i < NB_WND; ++i) wnd_bnd[i + 1] = wnd_bnd[i];

    /* derive_forward_lut */
    for (int val = 0; val < 1024; ++val) {
        int w = val >> LOG2_WND_RNG;
        int nb_step = val - (w << LOG2_WND_RNG);
        out->fwd_lut[val] = (uint16_t)clip_bd((int16_t)wnd_bnd[w] + (((int32_t)fwd_step[w] * nb_step + (1 << (LMCS_PREC - 1))) >> LMCS_PREC));
    }
    /* derive_backward_lut, get_bwd_idx */
    for (int val = 0; val < 1024; ++val) {
        int w = min_idx;
        for (; w < max_idx_plus1; ++w)
            if (val < wnd_bnd[w + 1]) break;
        if (w > NB_WND - 1) w = NB_WND - 1;
        int nb_step = val - wnd_bnd[w];
        out->bwd_lut[val] = (uint16_t)clip_bd((w << LOG2_WND_RNG) + (((int32_t)bwd_step[w] * nb_step + (1 << (LMCS_PREC - 1))) >> LMCS_PREC));
    }
    memcpy(out->wnd_bnd, wnd_bnd, sizeof(out->wnd_bnd));
    out->min_idx = (uint8_t)min_idx;
    out->max_idx = (uint8_t)max_idx_plus1;
    out->crs_offset = data->crs_offset;
    out->pad = 0;
    return OVHIP_OK;
}
