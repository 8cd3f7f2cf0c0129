// Third translation unit of hn_field2_hand.hip: the evaluation kernels of HN_PREC_F16 (k_field2_hand<MODE, 1>: hidden layers
// on one f16 MFMA per product instead of three), compiled beside the f16x3 kernels.
#define HN_HAND_F16_TU 1
#include "hn_field2_hand.hip"
