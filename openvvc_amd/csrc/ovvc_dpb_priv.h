/* ovvc_dpb_priv.h -- library-internal links between the DPB state machine (ovvc_dpb.c), its HIP back-end (ovvc_dpb_hip.hip)
 * and the stream driver (ovvc_stream.c). */
#ifndef OVVC_DPB_PRIV_H
#define OVVC_DPB_PRIV_H
#include "ovvc_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
int  ovhip_dpb_hip_ops_(const int *devices, int n_devices, ovhip_dpb_ops *ops, void **user);
void ovhip_dpb_hip_ops_free_(void *user);
void ovhip_dpb_rearm_(ovhip_dpb *d);
/* CLOCK_MONOTONIC seconds at which the frame's last picture was complete on the device / published (the stream driver's timeline) */
double ovhip_frame_published_at(const ovhip_frame *f);
double ovhip_frame_done_at(const ovhip_frame *f);
#ifdef __cplusplus
}
#endif
#endif
