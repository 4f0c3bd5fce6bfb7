/* vvc_dbf_tables.h -- H.266 clause 8.8.3.6.2, Table 43: deblocking thresholds tC' (indexed by
 * Q = Clip3(0, 65, qP + 2*(bS-1) + tc_offset) ... here 0..66 as the reference sizes it) and beta'
 * (indexed by Q = Clip3(0, 63, qP + beta_offset)); 10-bit: tc = tC', beta = beta' << 2
 * (libovvc/rcn_df.c:52-75, :171-188). */
#ifndef OVVC_VVC_DBF_TABLES_H
#define OVVC_VVC_DBF_TABLES_H
#include <stdint.h>
#ifndef OVT_ATTR
#define OVT_ATTR
#endif
OVT_ATTR static const uint16_t ovt_dbf_tc[67] = {
      0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,   0,
      3,   4,   4,   4,   4,   5,   5,   5,   5,   7,   7,   8,   9,  10,  10,  11,
     13,  14,  15,  17,  19,  21,  24,  25,  29,  33,  36,  41,  45,  51,  57,  64,
     71,  80,  89, 100, 112, 125, 141, 157, 177, 198, 222, 250, 280, 314, 352, 395,
      0   /* index 66: the reference sizes tc_lut[67] with 66 initialisers (rcn_df.c:52-63); kept */
};
OVT_ATTR static const uint8_t ovt_dbf_beta[65] = {
     0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,
     6,  7,  8,  9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 22, 24,
    26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56,
    58, 60, 62, 64, 66, 68, 70, 72, 74, 76, 78, 80, 82, 84, 86, 88,
     0   /* index 64: beta_lut[65] has 64 initialisers in the reference (rcn_df.c:65-75); kept */
};
#endif
