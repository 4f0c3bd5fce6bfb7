/* vvc_mc_taps.h -- H.266 fractional-sample interpolation filter coefficients
 * (clause 8.5.6.3, Tables 27/28/30/33), written for this repo's unified separable kernel:
 * row 0 is the integer-position identity filter so that one code path covers copy / H / V / HV
 * (the reference keeps 15-row tables indexed by frac-1 and four functions per block class:
 * libovvc/rcn_mc.c:67-142, :382-533).  Row 16 of the luma table is the 6-tap half-sample
 * smoothing filter selected by AMVR half-pel precision (rcn_inter.c:572-577).
 * All rows sum to 64.
 */
#ifndef OVVC_VVC_MC_TAPS_H
#define OVVC_VVC_MC_TAPS_H
#include <stdint.h>
#ifndef OVT_ATTR
#define OVT_ATTR
#endif
#ifndef OVT_CONST
#ifdef __cplusplus
#define OVT_CONST static constexpr /* the device derives its packed tap tables at compile time */
#else
#define OVT_CONST static const
#endif
#endif

/* luma, 8 taps at integer offsets -3..+4, index = 1/16 fraction (16 = half-pel smoothing) */
OVT_ATTR OVT_CONST int8_t ovt_mc_luma[17][8] = {
    {  0, 0,   0, 64,  0,   0, 0,  0 },
    {  0, 1,  -3, 63,  4,  -2, 1,  0 },
    { -1, 2,  -5, 62,  8,  -3, 1,  0 },
    { -1, 3,  -8, 60, 13,  -4, 1,  0 },
    { -1, 4, -10, 58, 17,  -5, 1,  0 },
    { -1, 4, -11, 52, 26,  -8, 3, -1 },
    { -1, 3,  -9, 47, 31, -10, 4, -1 },
    { -1, 4, -11, 45, 34, -10, 4, -1 },
    { -1, 4, -11, 40, 40, -11, 4, -1 },
    { -1, 4, -10, 34, 45, -11, 4, -1 },
    { -1, 4, -10, 31, 47,  -9, 3, -1 },
    { -1, 3,  -8, 26, 52, -11, 4, -1 },
    {  0, 1,  -5, 17, 58, -10, 4, -1 },
    {  0, 1,  -4, 13, 60,  -8, 3, -1 },
    {  0, 1,  -3,  8, 62,  -5, 2, -1 },
    {  0, 1,  -2,  4, 63,  -3, 1,  0 },
    {  0, 3,   9, 20, 20,   9, 3,  0 },
};

/* luma 4x4 blocks (affine sub-blocks): 6-tap variants */
OVT_ATTR OVT_CONST int8_t ovt_mc_luma4[16][8] = {
    { 0, 0,   0, 64,  0,   0, 0, 0 },
    { 0, 1,  -3, 63,  4,  -2, 1, 0 },
    { 0, 1,  -5, 62,  8,  -3, 1, 0 },
    { 0, 2,  -8, 60, 13,  -4, 1, 0 },
    { 0, 3, -10, 58, 17,  -5, 1, 0 },
    { 0, 3, -11, 52, 26,  -8, 2, 0 },
    { 0, 2,  -9, 47, 31, -10, 3, 0 },
    { 0, 3, -11, 45, 34, -10, 3, 0 },
    { 0, 3, -11, 40, 40, -11, 3, 0 },
    { 0, 3, -10, 34, 45, -11, 3, 0 },
    { 0, 3, -10, 31, 47,  -9, 2, 0 },
    { 0, 2,  -8, 26, 52, -11, 3, 0 },
    { 0, 1,  -5, 17, 58, -10, 3, 0 },
    { 0, 1,  -4, 13, 60,  -8, 2, 0 },
    { 0, 1,  -3,  8, 62,  -5, 1, 0 },
    { 0, 1,  -2,  4, 63,  -3, 1, 0 },
};

/* chroma, 4 taps at offsets -1..+2, index = 1/32 fraction */
OVT_ATTR OVT_CONST int8_t ovt_mc_chroma[32][4] = {
    {  0, 64,  0,  0 }, { -1, 63,  2,  0 }, { -2, 62,  4,  0 }, { -2, 60,  7, -1 },
    { -2, 58, 10, -2 }, { -3, 57, 12, -2 }, { -4, 56, 14, -2 }, { -4, 55, 15, -2 },
    { -4, 54, 16, -2 }, { -5, 53, 18, -2 }, { -6, 52, 20, -2 }, { -6, 49, 24, -3 },
    { -6, 46, 28, -4 }, { -5, 44, 29, -4 }, { -4, 42, 30, -4 }, { -4, 39, 33, -4 },
    { -4, 36, 36, -4 }, { -4, 33, 39, -4 }, { -4, 30, 42, -4 }, { -4, 29, 44, -5 },
    { -4, 28, 46, -6 }, { -3, 24, 49, -6 }, { -2, 20, 52, -6 }, { -2, 18, 53, -5 },
    { -2, 16, 54, -4 }, { -2, 15, 55, -4 }, { -2, 14, 56, -4 }, { -2, 12, 57, -3 },
    { -2, 10, 58, -2 }, { -1,  7, 60, -2 }, {  0,  4, 62, -2 }, {  0,  2, 63, -1 },
};

#endif
