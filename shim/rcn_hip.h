/* rcn_hip.h -- the MI355X back-end's override block for OpenVVC's reconstruction dispatch table.
 *
 * This is the file a maintainer adds to the reference tree: it is compiled WITH the reference's own
 * headers (ctudec.h, rcn_structures.h, ...) and linked against libovvc_hip.so.  It replaces the orchestrator slots of
 * struct RCNFunctions (libovvc/rcn_structures.h:499-694) -- the same boundary the SSE4/AVX2/NEON back-ends bind to
 * (rcn.c:214-299) -- with recorders, and the last alf.rcn_alf_filter_line of a picture with the device flush.
 */
#ifndef OVVC_RCN_HIP_H
#define OVVC_RCN_HIP_H

#include <stddef.h>
#include <stdint.h>

struct RCNFunctions;
struct OVCTUDec;
struct ovhip_recorder;

/* Same signature and call site as rcn_init_functions() (libovvc/rcn.c:147-149): called right after the scalar fill
 * (rcn.c:172) on the table embedded in an OVCTUDec (ctudec.h:651).  bitdepth != 10 leaves the table untouched, like
 * the x86 SIMD path (rcn.c:217). */
void rcn_init_functions_hip(struct RCNFunctions *rcn_func, uint8_t ict_type, uint8_t lm_chroma_enabled,
                            uint8_t sps_chroma_vertical_collocated_flag, uint8_t lmcs_flag, uint8_t bitdepth);

/* ---- management (the table has no user pointer: state lives in a side table keyed by the OVCTUDec) ---- */

/* Record-only mode: bind a caller-owned recorder for a pic_w x pic_h picture instead of a device job.  The slots then
 * record, nothing is launched (command-stream capture for tests and offline replay).  Without this call the first
 * rcn_attach_frame_buff creates the device context + job and fails loudly when no HIP device is present. */
int  ovhip_shim_bind_recorder(const struct OVCTUDec *ctudec, struct ovhip_recorder *rec, int pic_w, int pic_h);
struct ovhip_recorder *ovhip_shim_recorder(const struct OVCTUDec *ctudec);
/* The reference-picture table of the current picture (OVPicture pointers, order of first use = the indices the recorded
 * units carry in ref0 / ref1). */
int  ovhip_shim_ref_pictures(const struct OVCTUDec *ctudec, const void **out, int cap);
/* Close the prediction calls still being collected into one CU (affine sub-blocks, BDOF blocks): done implicitly by
 * every other slot, needed explicitly only before reading the recorder directly. */
void ovhip_shim_flush_pending(struct OVCTUDec *ctudec);
/* First error latched since the picture began (0 = none; negative OVHIP_E*): the slots return void. */
int  ovhip_shim_last_error(const struct OVCTUDec *ctudec);
/* Entries of the picture's collocated motion planes (ovhip_job_tmvp_cells(): 4 per refined unit, derived on the device from the
 * units and the refined vectors) written where the reference's caller + tmvp_store_mv put what rcn_dmvr_mv_refine returned
 * (vcl_coding_unit.c:2629-2645; drv_lines.c:270-330).  Called by the shim's own row-end hooks; exposed for the record-only
 * harness. */
struct ovhip_tmvp_cell;
int  ovhip_shim_apply_tmvp_cells(struct OVCTUDec *ctudec, const struct ovhip_tmvp_cell *cells, size_t n_entries);
/* Picture-level side information the filter slots collected (valid until the next picture begins). */
struct ovhip_sao_ctu;
struct ovhip_alf_ctu;
struct ovhip_lmcs_luts;
const struct ovhip_sao_ctu *ovhip_shim_sao_params(const struct OVCTUDec *ctudec, size_t *n_ctu);
const struct ovhip_alf_ctu *ovhip_shim_alf_params(const struct OVCTUDec *ctudec, size_t *n_ctu);
const int16_t *ovhip_shim_alf_table(const struct OVCTUDec *ctudec, int which /* 0 luma coeff, 1 luma clip, 2 chroma coeff, 3 chroma clip, 4 cc */, size_t *n);
const struct ovhip_lmcs_luts *ovhip_shim_lmcs(const struct OVCTUDec *ctudec);
void ovhip_shim_release(const struct OVCTUDec *ctudec);
/* Where a frame thread's time goes inside the back-end: wall seconds this OVCTUDec spent in the installed hooks since the profile was
 * switched on (or last reset), the part of them in the device half (picture begin, eager DMVR rows, the submit: waits for reference
 * pictures, launches, ovhip_job_wait, the copy into the OVFrame), and the number of outermost hook calls.  Recording = the difference.
 * Off (default): one branch per hook. */
typedef struct ovhip_shim_profile {
    double seconds_in_hooks, seconds_device; uint64_t n_calls;
    double seconds_overhead_per_call;      /* what the profile's own bracket adds to seconds_in_hooks per call (two time-stamp reads), measured */
} ovhip_shim_profile;
void ovhip_shim_set_profile(int on);
int  ovhip_shim_get_profile(const struct OVCTUDec *ctudec, ovhip_shim_profile *out, int reset);
/* What the last alf.rcn_alf_filter_line of a picture copies into the OVFrame after the picture is complete: OVHIP_OUT_PLANES
 * (default: an unmodified application reads the frame there, dectest.c:372-409) or OVHIP_OUT_NONE (the application takes its
 * frames through ovhip_shim_frame_output / _digest: no 24.9 MB copy per 4K picture).  Also: environment OVVC_HIP_OUTPUT=none. */
void ovhip_shim_set_output(int mode);
/* Band-wise submission (ovhip_frame_band): CTU rows per band; 0 = every picture goes to the device at its end (ovhip_frame_submit).
 * Default 0 (whole pictures: faster on the live decoder at every band size); environment OVVC_HIP_BANDS.  Pictures cut into
 * rect entries are always submitted whole. */
void ovhip_shim_set_bands(int ctu_rows_per_band);
void ovhip_shim_band_stats(const struct OVCTUDec *ctudec, uint32_t *bands_sent, uint32_t *bands_deferred);
struct ovhip_dpb;
void ovhip_shim_set_dpb(struct ovhip_dpb *dpb);          /* the application's device DPB instead of the shim's own */
/* Optional hook for ovframe_unref() reaching zero: the frame's device picture returns to the pool at once (otherwise when the
 * frame pointer comes back for a new picture). */
struct Frame;
void ovhip_shim_frame_released(const struct Frame *frame);

/* ---- output path (replaces the plane-by-plane copy-out of examples/dectest.c:372-409) ---- */
struct Frame;
/* Bytes of `frame` cropped to its output_window, in the layout write_decoded_frame_to_file writes. */
size_t ovhip_shim_frame_bytes(const struct Frame *frame);
/* The decoded frame's device picture, cropped and packed on the device, into dst (ovhip_shim_frame_bytes() bytes). */
int  ovhip_shim_frame_output(const struct OVCTUDec *ctudec, const struct Frame *frame, void *dst);
/* MD5 over the per-row MD5 digests of the cropped frame (include/ovvc_hip.h, "Output path"): only 16 bytes leave the device. */
int  ovhip_shim_frame_digest(const struct OVCTUDec *ctudec, const struct Frame *frame, uint8_t out[16]);

#endif
