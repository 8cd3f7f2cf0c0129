// The adjoint instantiation of the hand field kernel (k_field2_hand<2>, body in hn_field2_hand_adj.inl) as its own
// translation unit, so that it compiles beside the evaluation kernels of hn_field2_hand.hip.
#define HN_HAND_ADJ_TU 1
#include "hn_field2_hand.hip"
