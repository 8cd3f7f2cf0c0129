// v2 hand field (f16x3 MFMA, LDS-streamed weights): per-bone local coordinates (anerf_emb_point) ->
// 1386-wide masked encoding -> SDFNetwork forward, analytic d sdf / d p (reverse sweep + encoding
// Jacobian) and RenderingNetwork, fused; 4 waves x 32 samples per workgroup, one workgroup per CU
// (hn_mlp2.h).  The 1386 features are generated once per sample tile as MFMA B fragments (fp16 hi/lo),
// parked in the wave's stash and streamed back for the three layers that consume them (lin0, the lin4
// skip columns, colour lin0) and for the Jacobian.
//
// Reference: utils/fields.py:22-52 (bone coordinates), :132-177 (sdf net, .gradient), :222-240 (colour
// net), called from utils/renderer.py:137-142 / 390-396.  The chunk order below is the contract with
// hn_pack2.hip (build_hand_stream); the feature k-slot layout is bone_slots / left_slots there.
#include <stdlib.h>

#include <atomic>

#include "hn_common.h"
#if !defined(HN_HAND_ADJ_TU) && !defined(HN_HAND_F16_TU) && !defined(HN_HAND_QUAD_TU) && !defined(HN_MFMA16)
#define HN_MFMA16 HN_HAND_EVAL_MFMA16   // the evaluation kernels (MODE 0, 1); the adjoint translation unit keeps 32x32x16
#endif
#include "hn_mlp2.h"
#ifndef HN_FEAT_ACQUIRE_VISIBLE
#define HN_FEAT_ACQUIRE_VISIBLE true
#endif
#ifndef HN_FEAT_LOADS_BEHIND
#define HN_FEAT_LOADS_BEHIND 1
#endif
#ifndef HN_JAC_VARIANT
#define HN_JAC_VARIANT 2
#endif
#ifndef HN_PARK_AGPR
#define HN_PARK_AGPR 1
#endif
#ifndef HN_EXP_ACC
#define HN_EXP_ACC 4
#endif
#ifndef HN_DEFER_STASH
#define HN_DEFER_STASH 0   // 1: the fp32 activation tiles of the hidden layers go to the stash one chunk late (to_regs_hold): measured, no gain
#endif

namespace hn {
namespace v2 {

__constant__ float c_cutoff2[N_BONES] = {0.08f, 0.03f, 0.03f, 0.02f, 0.02f, 0.03f, 0.02f, 0.02f, 0.02f, 0.03f, 0.02f,
                                         0.02f, 0.02f, 0.03f, 0.02f, 0.02f, 0.02f, 0.03f, 0.02f, 0.02f, 0.02f};
constexpr float TAU2 = 200.f;
// Samples the adjoint DROPPED (g_pts = 0, no share in the pose gradients, zero rows in the parameter-gradient signals) because their
// adjoint quantities left the fp16 fragments' range -- within ~2 mm of a bone's origin, hn_field2_hand_adj.inl.  Counted per device since
// the library was loaded (hn_dropped_samples).  Only the adjoint translation unit's copy is ever written or read.
static __device__ unsigned long long g_hn_dropped_samples = 0ull;

struct Hand2Args {
    const float* pts;      // [n,3]
    const float* bt_inv;   // [n_frames,21,4,4]
    const float* T_pose;   // [n_frames,21,3]
    int n_pts;
    int pts_per_frame;
    int n_frames;
    const char* blob;
    size_t blob_bytes;
    const float* b8;       // &bias of lin8's sdf row (the field's retained copy: read on the device, a re-pack waits for nothing)
    const float* c_blast;  // the three biases of colour lin4 (the field's retained copy, read on the device)
    float* sdf;
    float* grad;
    float* rgb;
    float* feat;
    float4* scratch;
    int dbg;
    int cull;   // hn_field_set_culling
    // adjoint (MODE 2): upstream gradients in, input / pose gradients out
    const float* g_sdf;    // [n]
    const float* g_grad;   // [n,3]
    const float* g_rgb;    // [n,3]
    float* g_pts;          // [n,3]
    float* g_bt_inv;       // [n_frames,21,4,4] accumulated (atomics), or NULL
    float* g_T_pose;       // [n_frames,21,3] accumulated (atomics), or NULL
    unsigned* xsync;       // 16 zeroed counters (8 XCDs x {members, arrivals}) or NULL: see "XCD pacing" in the kernel
    const int* n_pts_dev;  // NULL, or the sample count on the DEVICE (<= n_pts): a compacted list whose length the host does not know
    const int* orig_idx;   // NULL, or per sample of a compacted list its index in the dense list (what the frame is taken from)
    float* pose_part;      // adjoint: NULL, or [n_tiles * 4][2][21 * 12] -- per sample TILE and wave the sums of the pose-gradient addends of
                           // the wave's first frame [0] and of the frame behind it [1] (a wave of a dense multi-frame list may cross one
                           // frame boundary); k_pose_part_reduce adds a frame's rows in a fixed order that depends on that frame's own
                           // tiles only: the same bits in every run AND whatever other frames share the launch.  Atomics otherwise.
    const int* frame_seg;  // NULL, or the frame table of a frame-aligned compact list (hn_common.h: launch_frame_seg)
    // MODE 5 (the adjoint from a tape that also leaves the PER-LAYER SIGNALS of the parameter gradients): HSG_COUNT row-major
    // [n, 256] fp32 arrays `sig + k * sig_pitch` (enum below; the layout of hn_field2_obj.hip's), unscaled, and gb [n,3] (the adjoint of
    // d sdf / d pts incl. the colour network's share: J gb is the forward-direction sweep's input)
    float* sig;
    size_t sig_pitch;      // floats between two signal arrays
    float* gb_out;
};
//   HSG_CB + k: adjoint of colour layer (3 - k)'s pre-activation;  HSG_C + k: c_{k+1};  HSG_A + l: a_{l+1};  HSG_DZ + l: dz_l of the reverse
//   sweep;  HSG_V + l: v_l = sigma'_l dzb_l (forward-direction sweep);  HSG_ZB + l: zb_l (second reverse sweep);  HSG_FB: the feature vector's adjoint
enum { HSG_CB = 0, HSG_C = 4, HSG_A = 8, HSG_DZ = 16, HSG_V = 24, HSG_ZB = 32, HSG_FB = 40, HSG_COUNT = 41 };

// stash slots of one wave (32 KiB each)
enum {
    HS_A1 = 0,      // a1..a7 -> 0..6 (fp32 activations for the reverse sweep)
    HS_DZ7 = 7,     // fragments
    HS_FVEC = 8,    // fragments
    HS_DZ4 = 9,     // fragments
    HS_A4F = 10,    // (free: a4 used to pass through here as fragments)
    HS_FEAT = 11,   // 87 k-step blocks of feature fragments (84 bone + 3 leftover) = 174 KiB -> 6 slots
    HS_LEFT = 17,   // 21 x 64 floats: the leftover (r_1 | r_2) h values while the bones are generated
    HAND2_SLOTS = 18,
    HAND2_SLOTS_SDF = 18,
    // ... and what the adjoint (MODE 2) adds
    HS_A8 = 18,     // a8 (fp32)
    HS_DZ = 19,     // dz0..dz7 of the reverse sweep as fp32 tiles -> 19..26; slot l is overwritten by the second-order
                    // source w_l once the forward-direction sweep has passed layer l
    HS_C = 27,      // colour activations c1..c4 -> 27..30 (relu masks)
    HS_GXB = 31,    // J gb as 87 k-step blocks of fragments (the layout of HS_FEAT) -> 31..36
    HS_TS = 37,     // per bone 9 sums of d sdf / d features (T0, T1[4], T2[4]), one float per lane each -> 37, 38
    HS_QA = 39,     // per bone the colour network's share of the bone-frame gradient qbar (3 floats per lane);
                    // behind it (+ 16 KiB) the leftover rows of the X adjoint, one float per bone and lane
    HS_ZB4 = 40,    // zb4 as fragments
    HAND2_SLOTS_ADJ = 41,
};
constexpr int FEAT_BLOCKS = 4 * N_BONES;     // first leftover block index
constexpr int STAGE_BYTES = 8 * 1024;        // LDS staging of one bone's 4 fragment pairs (Jacobian pass)
constexpr int POSE_ROW = 256;                // floats per wave and frame: 21 bones x 12 pose-gradient addends (Hand2Args::pose_part)
constexpr int POSE_FRAMES = 2;               // frames a WAVE may touch for the atomics-free pose gradients: its first sample's and the next
constexpr int POSE_MAX_TILES = 8192;         // tiles of a launch up to which the rows are kept (8 KiB per tile); beyond: atomics
constexpr size_t HAND2_LDS_POSE = 2 * CHUNK_MAX + WG_WAVES * STAGE_BYTES + 16;   // (+ 16: the 4 per-wave bone masks of the culling)

constexpr int HB_HID = chunk_bytes(1, 16, true);
constexpr int HB_BWD = chunk_bytes(1, 16, false);
constexpr int HB_BONE = chunk_bytes(4, 4, false);
constexpr int KS_LEFT = S16 ? 4 : 3;                  // k-steps of the leftover block (16x16x32: pairs, the fourth is empty)
constexpr int HB_LEFT_T = chunk_bytes(4, KS_LEFT, true);    // leftover chunk with the 4 biases (lin0)
constexpr int HB_LEFT = chunk_bytes(4, KS_LEFT, false);
constexpr int HB_G = chunk_bytes(4, 2, true);         // colour lin0: enc(g) columns + the 4 biases
constexpr int HB_W4ROWS = 3 * TAIL_BYTES;             // adjoint: the three rows of colour lin4, one tail-format KiB each

struct Bone2 {
    float v, r[3], hh;
};
// utils/fields.py:26-35: q = R_b p + t_b - T_b; v = |q|; r = q / v; h = 1 - sigmoid(200 (v - cutoff_b))
// Element i of a read-only array.  UNI: the pointer is wave-uniform (all 64 samples belong to one frame, the usual
// case) and the load goes through the constant address space, i.e. the scalar data cache: no vector-memory round trip
// and no VGPRs for the 15 pose values a bone needs.
typedef const float __attribute__((address_space(4))) cfloat_t;
template <bool UNI>
__device__ __forceinline__ float rd(const float* p, int i) {
    if constexpr (UNI)
        return ((cfloat_t*)p)[i];
    else
        return p[i];
}
template <bool UNI>
__device__ __forceinline__ Bone2 bone_coords2(const float p[3], const float* M, const float* Tp, int b) {
    Bone2 o;
    const float* m = M + 16 * b;
    const float q0 = rd<UNI>(m, 0) * p[0] + rd<UNI>(m, 1) * p[1] + rd<UNI>(m, 2) * p[2] + rd<UNI>(m, 3) - rd<UNI>(Tp, 3 * b);
    const float q1 = rd<UNI>(m, 4) * p[0] + rd<UNI>(m, 5) * p[1] + rd<UNI>(m, 6) * p[2] + rd<UNI>(m, 7) - rd<UNI>(Tp, 3 * b + 1);
    const float q2 = rd<UNI>(m, 8) * p[0] + rd<UNI>(m, 9) * p[1] + rd<UNI>(m, 10) * p[2] + rd<UNI>(m, 11) - rd<UNI>(Tp, 3 * b + 2);
    o.v = sqrtf(q0 * q0 + q1 * q1 + q2 * q2);
    o.r[0] = q0 / o.v;   // no epsilon: a sample on a joint is NaN, as in the reference (SURVEY B-10)
    o.r[1] = q1 / o.v;
    o.r[2] = q2 / o.v;
    const float sg = 1.f / (1.f + expf(-TAU2 * (o.v - c_cutoff2[b])));
    o.hh = 1.f - sg;
    return o;
}
__device__ __forceinline__ float sc_half2(float ang, int h) {
    float s, c;
    sincos_cw(ang, s, c);
    return h ? c : s;
}
// the 4 k-steps x 8 values of one bone for this lane (bone_slots in hn_pack2.hip), already times h
// The two lanes of a sample need the sine (half 0) and the cosine (half 1) of the same 31 angles.  Each computes the pair
// for HALF of them -- half 0 the angles of k-steps 0 and 1, half 1 those of k-steps 2 and 3 -- keeps the member it needs
// and hands the other one to its partner (v_permlane32_swap): 16 sincos per lane and bone instead of 32, the same values
// bit for bit (the same routine on the same angle).  Feature generation runs with the matrix pipe idle (138 000 cycles per
// tile in the in-kernel stamps at full load).
__device__ __forceinline__ void bone_features2(const Bone2& bn, int h, float (&f)[4][8]) {
    // angle of slot (s, j), s < 2 for half 0 / slot (s + 2, j) for half 1
    float keep[2][8], give[2][8];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // slot (s, j):     s = 0: v 2^j;  s = 1: j < 2: v 2^(8+j), else r_0 2^(j-2)
            // slot (s + 2, j): s = 0: j < 1: r_0 64, else r_1 2^(j-1);  s = 1: r_2 2^j (j < 7; (3, 7) is the raw pair)
            const float a_lo = s == 0 ? bn.v * (float)(1 << j) : (j < 2 ? bn.v * (float)(256 << j) : bn.r[0] * (float)(1 << (j - 2)));
            const float a_hi = s == 0 ? (j < 1 ? bn.r[0] * 64.f : bn.r[1] * (float)(1 << (j - 1))) : bn.r[2] * (float)(1 << (j & 7));
            float sn, cs;
            sincos_cw(h ? a_hi : a_lo, sn, cs);
            keep[s][j] = h ? cs : sn;    // this half's function of its own angle
            give[s][j] = h ? sn : cs;    // the partner's function of that angle
        }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float got = other_half(give[s][j], h);   // half 0 receives sin of slot (s + 2, j), half 1 cos of slot (s, j)
            f[s][j] = (h ? got : keep[s][j]) * bn.hh;
            f[s + 2][j] = (h ? keep[s][j] : got) * bn.hh;
        }
    f[3][7] = (h ? bn.r[0] : bn.v) * bn.hh;
}
__device__ __forceinline__ void encode_v4h(const float v[3], int h, float (&f)[2][8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) f[0][j] = sc_half2(v[j >> 2] * (float)(1 << (j & 3)), h);
#pragma unroll
    for (int j = 0; j < 4; ++j) f[1][j] = sc_half2(v[2] * (float)(1 << j), h);
    f[1][4] = h ? v[2] : v[0];
    f[1][5] = h ? 0.f : v[1];
    f[1][6] = 0.f;
    f[1][7] = 0.f;
}

// Encoding Jacobian of one bone's 64 main slots.  own[s][j] = this lane's stored feature (phi * h); its
// partner lane (other half, same sample) stores the conjugate function (cos for sin) * h, so
//   d(sin(f x) h)/dx = +f partner,  d(cos(f x) h)/dx = -f partner,  and every slot adds own * h'/h to d/dv
// with h'/h = -tau sigmoid(tau (v - cutoff)) =: kk.  G0 covers k-steps 0,1 (register 8 s + j), G1 k-steps 2,3.
__device__ __forceinline__ void bone_jacobian(const f32x16& G0, const f32x16& G1, const float (&own)[4][8], const Bone2& bn,
                                              float kk, int h, float& Sv, float (&Sr)[3]) {
    float sum_own = 0.f;   // sum G * own  (the h'/h term)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float Gv = (s < 2) ? G0[8 * s + j] : G1[8 * (s - 2) + j];
            sum_own = fmaf(Gv, own[s][j], sum_own);
            if (s == 3 && j == 7) {
                // raw pair (v | r_0) h
                Sv += h ? 0.f : Gv * bn.hh;
                Sr[0] += h ? Gv * bn.hh : 0.f;
                continue;
            }
            const float other = other_half(own[s][j], h);
            // which variable / frequency this slot encodes
            int var, k;   // var 0 = v, 1..3 = r_0..r_2
            if (s == 0) {
                var = 0;
                k = j;
            } else if (s == 1) {
                var = j < 2 ? 0 : 1;
                k = j < 2 ? 8 + j : j - 2;
            } else if (s == 2) {
                var = j < 1 ? 1 : 2;
                k = j < 1 ? 6 : j - 1;
            } else {
                var = 3;
                k = j;
            }
            const float fr = (float)(1 << k);
            const float t = Gv * (h ? -fr : fr) * other;
            if (var == 0)
                Sv += t;
            else
                Sr[var - 1] += t;
        }
    Sv = fmaf(sum_own, kk, Sv);
}
// (Sv, Sr) of one bone -> d sdf / d p contribution: d/dq = Sv r + (Sr - (Sr.r) r) / v ; d/dp = R_b^T d/dq
template <bool UNI>
__device__ __forceinline__ void bone_to_p(float Sv, const float (&Sr)[3], const Bone2& q, const float* m, float (&g)[3]) {
    const float sv = half_sum(Sv);
    const float sr0 = half_sum(Sr[0]), sr1 = half_sum(Sr[1]), sr2 = half_sum(Sr[2]);
    const float dot = sr0 * q.r[0] + sr1 * q.r[1] + sr2 * q.r[2];
    const float dq0 = sv * q.r[0] + (sr0 - dot * q.r[0]) / q.v;
    const float dq1 = sv * q.r[1] + (sr1 - dot * q.r[1]) / q.v;
    const float dq2 = sv * q.r[2] + (sr2 - dot * q.r[2]) / q.v;
    g[0] += rd<UNI>(m, 0) * dq0 + rd<UNI>(m, 4) * dq1 + rd<UNI>(m, 8) * dq2;
    g[1] += rd<UNI>(m, 1) * dq0 + rd<UNI>(m, 5) * dq1 + rd<UNI>(m, 9) * dq2;
    g[2] += rd<UNI>(m, 2) * dq0 + rd<UNI>(m, 6) * dq1 + rd<UNI>(m, 10) * dq2;
}


// ---- bone-frame calculus of the adjoint (oracle/field_bwd.py::_HandInput) -------------------------------------------
// A bone's 66 features are phi_f(y_f) h(v) with y_f one of (v, r_0, r_1, r_2).  For a row G over the features the
// h-weighted sums
//   T0 = sum_f G_f phi_f h,   T1_a = sum_{f on a} G_f phi_f' h,   T2_a = sum_{f on a} G_f phi_f'' h
// give everything the input map needs without dividing by h: with kk = h'/h = -tau (1 - h) and
// k2 = h''/h = -tau^2 (1 - h)(2 h - 1):  Sv = T1_v + kk T0,  Sr = T1_r,  d/dq = Sv r + (Sr - (Sr.r) r) / v.
// The stored feature of a slot is own = phi h; its partner lane (other half, same sample) stores the conjugate
// function, so phi' h = +-f partner and phi'' h = -f^2 own.
struct BoneSums {
    float T0, T1[4], T2[4];
};
// which argument / frequency the main slot (s, j) of a bone encodes (bone_slots in hn_pack2.hip); raw: the (v | r_0) pair
__device__ __forceinline__ constexpr int slot_var(int s, int j) { return s == 0 ? 0 : (s == 1 ? (j < 2 ? 0 : 1) : (s == 2 ? (j < 1 ? 1 : 2) : 3)); }
__device__ __forceinline__ constexpr int slot_freq(int s, int j) { return s == 0 ? j : (s == 1 ? (j < 2 ? 8 + j : j - 2) : (s == 2 ? (j < 1 ? 6 : j - 1) : j)); }
// per-lane partial sums over the 64 main slots (half_sum them afterwards); G0: k-steps 0, 1 (register 8 s + j), G1: 2, 3
template <bool WITH2>
__device__ __forceinline__ void bone_sums(const f32x16& G0, const f32x16& G1, const float (&own)[4][8], float hh, int h, BoneSums& S) {
    S.T0 = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) S.T1[a] = S.T2[a] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float Gv = (s < 2) ? G0[8 * s + j] : G1[8 * (s - 2) + j];
            S.T0 = fmaf(Gv, own[s][j], S.T0);
            if (s == 3 && j == 7) {   // raw pair (v | r_0) h: phi' = 1, phi'' = 0
                S.T1[0] += h ? 0.f : Gv * hh;
                S.T1[1] += h ? Gv * hh : 0.f;
                continue;
            }
            const int var = slot_var(s, j);
            const float fr = (float)(1 << slot_freq(s, j));
            const float other = other_half(own[s][j], h);
            S.T1[var] = fmaf(Gv, (h ? -fr : fr) * other, S.T1[var]);
            if (WITH2) S.T2[var] = fmaf(Gv, -(fr * fr) * own[s][j], S.T2[var]);
        }
}
__device__ __forceinline__ void sums_reduce(BoneSums& S, bool with2) {
    S.T0 = half_sum(S.T0);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        S.T1[a] = half_sum(S.T1[a]);
        if (with2) S.T2[a] = half_sum(S.T2[a]);
    }
}
// d/dq of F = sum_f G_f F_f from the (reduced) sums
__device__ __forceinline__ void dq_from_sums(const BoneSums& S, const Bone2& q, float kk, float (&dq)[3]) {
    const float Sv = fmaf(kk, S.T0, S.T1[0]);
    const float dot = S.T1[1] * q.r[0] + S.T1[2] * q.r[1] + S.T1[3] * q.r[2];
#pragma unroll
    for (int c = 0; c < 3; ++c) dq[c] = Sv * q.r[c] + (S.T1[1 + c] - dot * q.r[c]) / q.v;
}
// Hessian-vector product of F at q along w (the second-order term of g . gbar, _HandInput.second)
__device__ __forceinline__ void hv_from_sums(const BoneSums& S, const Bone2& q, float kk, float k2, const float (&w)[3], float (&hv)[3]) {
    const float rw = q.r[0] * w[0] + q.r[1] * w[1] + q.r[2] * w[2];
    float wt[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) wt[c] = (w[c] - q.r[c] * rw) / q.v;
    const float Sv = fmaf(kk, S.T0, S.T1[0]);
    const float Sr[3] = {S.T1[1], S.T1[2], S.T1[3]};
    const float dSv_dv = 2.f * kk * S.T1[0] + S.T2[0] + k2 * S.T0;
    const float hb_r = kk * (Sr[0] * q.r[0] + Sr[1] * q.r[1] + Sr[2] * q.r[2]);
    const float cc = kk * (Sr[0] * wt[0] + Sr[1] * wt[1] + Sr[2] * wt[2]);
    const float e[3] = {S.T2[1] * wt[0], S.T2[2] * wt[1], S.T2[3] * wt[2]};
    const float e_r = e[0] * q.r[0] + e[1] * q.r[1] + e[2] * q.r[2];
    const float dot = Sr[0] * q.r[0] + Sr[1] * q.r[1] + Sr[2] * q.r[2];
    const float sr_wt = Sr[0] * wt[0] + Sr[1] * wt[1] + Sr[2] * wt[2];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float gSv = dSv_dv * q.r[c] + (kk * Sr[c] - hb_r * q.r[c]) / q.v;
        hv[c] = gSv * rw + Sv * wt[c] + cc * q.r[c] + (e[c] - e_r * q.r[c]) / q.v -
                ((Sr[c] - dot * q.r[c]) / q.v * rw + dot * wt[c]) / q.v - sr_wt * q.r[c] / q.v;
    }
}
// sum over the wave's 64 lanes, every lane gets the total: four DPP steps inside the 16-lane rows (quad swaps, half-row and
// row mirrors: after each step the lanes of the group reached hold the group's sum), then the rows with the two lane-row
// swaps of gfx950.  All VALU: the LDS-crossbar butterfly it replaces (6 dependent ds_bpermute per value, 12 values per live
// bone) was most of the 12 900 cycles a bone of the adjoint's second pass took in the in-kernel stamps.
template <int CTRL>
__device__ __forceinline__ float dpp_xadd(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum64(float v) {
    v = dpp_xadd<0xB1>(v);    // quad_perm [1,0,3,2]
    v = dpp_xadd<0x4E>(v);    // quad_perm [2,3,0,1]
    v = dpp_xadd<0x141>(v);   // row_half_mirror
    v = dpp_xadd<0x140>(v);   // row_mirror
    unsigned x = __builtin_bit_cast(unsigned, v);
    auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    v = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
    x = __builtin_bit_cast(unsigned, v);
    r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}

// MODE 0: sdf only (sampling passes); 1: full evaluation (sdf, d sdf / d p, colour); 2: full evaluation followed by its
// adjoint (hn_field_eval_bwd): the sweeps of oracle/field_bwd.py in the same weight-stream / register-resident form,
// per sample tile, with the tape in the wave's stash.
// The fitting step splits mode 2 in two launches so that nothing is evaluated twice: 3 = full evaluation that keeps its
// tape (stash slots per sample TILE, in a buffer the caller keeps until the backward pass), 4 = the adjoint alone, from
// that tape.
// HP: MFMA passes per product of the hidden layers -- 3: fp32-equivalent (HN_PREC_F16X3); 1: HN_PREC_F16, the single-pass
// throughput mode (evaluation modes 0 / 1 only; which layer groups drop to one pass: the HN_F16_* macros below)
#ifndef HN_F16_FEAT_PASSES
#define HN_F16_FEAT_PASSES 3   // lin0 and the lin4 skip columns: the 1386 encoded features (sin / cos up to 2^9 v)
#endif
#ifndef HN_F16_JAC_PASSES
#define HN_F16_JAC_PASSES 3    // W0^T dz0 + W4x^T dz4 in front of the encoding Jacobian (entries are multiplied by up to 2^9)
#endif
#ifndef HN_F16_LIN7_PASSES
#define HN_F16_LIN7_PASSES 3   // the last hidden layer of the SDF network (-> a8 -> sdf, the seed of the reverse sweep)
#endif
#ifndef HN_F16_COLOR_PASSES
#define HN_F16_COLOR_PASSES 3  // the colour network (measured at one pass: rgb 3e-2 from the reference on the fixture; at three: see tests)
#endif
template <int MODE, int HP>
__device__ __forceinline__ void field2_hand_body(const Hand2Args& a) {
    static_assert(HP == 3 || MODE <= 1, "the single-pass mode exists for the evaluation kernels");
    constexpr int PH = HP;                                  // hidden layers, reverse sweep, colour network
    constexpr int PF = HP == 1 ? HN_F16_FEAT_PASSES : 3;
    constexpr int PJ = HP == 1 ? HN_F16_JAC_PASSES : 3;
    constexpr int P7 = HP == 1 ? HN_F16_LIN7_PASSES : 3;
    constexpr int PC = HP == 1 ? HN_F16_COLOR_PASSES : 3;
    using IPF = std::integral_constant<int, PF>;
    using IPC = std::integral_constant<int, PC>;
    using IP3 = std::integral_constant<int, 3>;
    constexpr bool FULL = MODE >= 1;
    constexpr bool ADJ = MODE >= 2;                    // the forward pass writes the tape
    constexpr bool RUN_FWD = MODE != 4 && MODE != 5;
    constexpr bool RUN_ADJ = MODE == 2 || MODE == 4 || MODE == 5;   // 5: 4 + the per-layer signals of the parameter gradients (Hand2Args::sig)
    [[maybe_unused]] constexpr bool PG = MODE == 5;
    constexpr bool PER_TILE = MODE >= 3;               // stash indexed by tile (kept across launches), not by workgroup
    constexpr int N_SLOTS = ADJ ? HAND2_SLOTS_ADJ : HAND2_SLOTS;
    constexpr int FIRST_CHUNK = MODE >= 4 ? HB_W4ROWS : HB_BONE;   // first chunk of a tile's program
    extern __shared__ __attribute__((aligned(16))) char lds[];
    f16_flush_mode();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31;
    const int h = lane >> 5;
#ifndef HN_ADJ_WB
#define HN_ADJ_WB 0   // (A/B, round 4: write-back stores for the adjoint kernel's w_l tiles too, in the hope that the second reverse sweep
                      //  finds them in L2 / MALL: k_field2_hand<4> 0.933 -> 0.971 ms.  nt stays.)
#endif
    StashT<(MODE <= 1 || (HN_ADJ_WB && MODE >= 4)) ? STASH_ST_AUX : STASH_AUX> sh;   // (evaluation kernels: write-back stores; taped / adjoint kernels: nt -- hn_mlp2.h)
    sh.init(a.scratch + ((size_t)blockIdx.x * WG_WAVES + wave) * N_SLOTS * SLOT_F4, N_SLOTS, lane);
    constexpr int FEAT = HS_FEAT * SLOT_BYTES;   // byte offset of the feature fragment blocks
    constexpr int LEFT = HS_LEFT * SLOT_BYTES;   // ... of the leftover values
    constexpr int GXB = HS_GXB * SLOT_BYTES;     // adjoint: J gb, same block layout as the features ...
    constexpr int LEFT2 = LEFT + 8192;           // ... and its leftover values
    constexpr int TS = HS_TS * SLOT_BYTES;       // per-bone sums of d sdf / d features
    constexpr int QA = HS_QA * SLOT_BYTES;       // colour network's share of qbar per bone (3 floats per lane)
    constexpr int LEFTX = QA + 16384;            // leftover rows of an X-space adjoint, one float per bone and lane
    constexpr int NZ_OFF = LEFT + 6144;          // the wave's live-bone mask `nz`, for the adjoint launch
    constexpr int LEFTJ = LEFT + 16384;          // the Jacobian pass's leftover rows, one float per bone and lane
    int feat_base = FEAT;                        // which block set load_bone reads (FEAT or GXB)
    const int n_pts = a.n_pts_dev != nullptr ? __builtin_amdgcn_readfirstlane(*a.n_pts_dev) : a.n_pts;
    const int n_tiles = (n_pts + WG_SAMPLES - 1) / WG_SAMPLES;

    WStream ws;
    ws.init(a.blob, a.blob_bytes, lds, wave, lane);
    char* const stage = lds + 2 * CHUNK_MAX + wave * STAGE_BYTES;   // per-wave staging of one bone's fragments
    ws.dbg_nofetch = (HN_DBG(a) & 4) ? 1 : 0;
    if ((int)blockIdx.x < n_tiles) ws.template fetch_all_c<FIRST_CHUNK>();

    // XCD pacing (hn_mlp2.h): the workgroups of an XCD meet at every tile start of a long launch.  (Further meeting points
    // inside the tile program were tried -- before the reverse sweep, before colour lin0 -- and lose 0.5 - 1 %.)
    XcdPace xp;
    xp.init(a.xsync);
    const int full_rounds = n_tiles / (int)gridDim.x;
    float* const prow = reinterpret_cast<float*>(lds + HAND2_LDS_POSE) + wave * (POSE_FRAMES * POSE_ROW);   // this wave's pose-gradient sums per frame (adjoint modes)
    for (int tile = blockIdx.x, it = 0; tile < n_tiles; tile += gridDim.x, ++it) {
        if constexpr (RUN_ADJ) {
            if (a.pose_part != nullptr) {   // this tile's rows start from zero (a wave's LDS operations complete in order)
                for (int i = lane; i < POSE_FRAMES * POSE_ROW; i += 64) prow[i] = 0.f;
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (xp.on() && it >= 1 && it < full_rounds && it % XCD_PACE_EVERY == 0) xp.meet(it / XCD_PACE_EVERY);
        ws.stamp(6);   // (timing builds: tile start)
        const bool more = tile + (int)gridDim.x < n_tiles;
        const int n = tile * WG_SAMPLES + wave * 32 + j;
        const bool valid = n < n_pts;
        const int nn = valid ? n : n_pts - 1;
        const float p[3] = {a.pts[3 * nn], a.pts[3 * nn + 1], a.pts[3 * nn + 2]};
        int dense_i = a.orig_idx != nullptr ? a.orig_idx[nn] : nn;
        dense_i = dense_i < 0 ? -1 - dense_i : dense_i;   // (a pad of the frame-aligned compact list: a copy of dense sample -1 - idx)
        int frame = dense_i / a.pts_per_frame;
        frame = frame < a.n_frames ? frame : a.n_frames - 1;
        const float* M = a.bt_inv + (size_t)frame * N_BONES * 16;
        const float* Tp = a.T_pose + (size_t)frame * N_BONES * 3;
        // the usual case: the wave's 32 samples belong to one frame -> pose values through the scalar cache
        const int frame0 = __builtin_amdgcn_readfirstlane(frame);
        const bool uni = __ballot(frame != frame0) == 0ull;
        const float* Mu = a.bt_inv + (size_t)frame0 * N_BONES * 16;
        const float* Tu = a.T_pose + (size_t)frame0 * N_BONES * 3;
        auto coords = [&](int b) { return uni ? bone_coords2<true>(p, Mu, Tu, b) : bone_coords2<false>(p, M, Tp, b); };
        auto to_p = [&](float Sv, const float(&Sr)[3], const Bone2& bn, int b, float(&g)[3]) {
            if (uni)
                bone_to_p<true>(Sv, Sr, bn, Mu + 16 * b, g);
            else
                bone_to_p<false>(Sv, Sr, bn, M + 16 * b, g);
        };

        if constexpr (PER_TILE) sh.init(a.scratch + ((size_t)tile * WG_WAVES + wave) * N_SLOTS * SLOT_F4, N_SLOTS, lane);
        // ---- F0: features of the 21 bones -> fragments in the stash -------------------------------------
        // nz bit b: some sample of this wave has a non-zero mask h for bone b.  Where none has, all 64
        // features of the bone are exactly 0 for the whole wave: nothing is generated or stored, and the
        // consumers substitute zero fragments instead of loading (the MFMAs still run: dense compute).
        ws.stamp(7);   // (points and frame known)
        unsigned nz = 0;
        if constexpr (!RUN_FWD) {
            nz = __builtin_bit_cast(unsigned, sh.f32_load(NZ_OFF));   // as the evaluation launch left it
        } else {
#pragma unroll 1
        for (int b = 0; b < N_BONES; ++b) {
            const Bone2 bn = coords(b);
            const bool any = __ballot(bn.hh != 0.f) != 0ull;
            if (any) {
                nz |= 1u << b;
                float f[4][8];
                bone_features2(bn, h, f);
                h8 fh[4], fl[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) split8(f[s], fh[s], fl[s]);
#pragma unroll
                for (int s = 0; s < 4; s += 2) {   // (16x16x32: k-step pairs -> the two column blocks' fragments)
                    frags_in(fh[s], fh[s + 1]);
                    frags_in(fl[s], fl[s + 1]);
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) sh.frag_store(FEAT, 4 * b + s, fh[s], fl[s]);
            }
            sh.f32_store(LEFT + b * 256, any ? (h ? bn.r[2] : bn.r[1]) * bn.hh : 0.f);
        }
        nz = __builtin_amdgcn_readfirstlane(nz);
        ws.stamp(8);   // (bone loop of the feature generation done)
        {   // leftover block: element j of k-step u belongs to bone 8u + j
            h8 fh[4], fl[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float f[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) f[jj] = (u < 3 && 8 * u + jj < N_BONES) ? sh.f32_load(LEFT + (8 * u + jj) * 256) : 0.f;
                split8(f, fh[u], fl[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; u += 2) {
                frags_in(fh[u], fh[u + 1]);
                frags_in(fl[u], fl[u + 1]);
            }
#pragma unroll
            for (int u = 0; u < KS_LEFT; ++u) sh.frag_store(FEAT, FEAT_BLOCKS + u, fh[u], fl[u]);
        }
        if constexpr (ADJ) sh.f32_store(NZ_OFF, __builtin_bit_cast(float, nz));
        }   // RUN_FWD
        nz = __builtin_amdgcn_readfirstlane(nz);

        // nzw: the bones whose weight chunks this workgroup runs.  Dense: all of them.  Culled: those that are live
        // in at least one of the 4 waves (the chunks are shared through LDS, so the skip has to be workgroup-wide),
        // plus bone 0, whose first chunk is always in flight when a pass starts.
        unsigned nzw = (1u << N_BONES) - 1u;
        if (a.cull) {
            unsigned* ex = reinterpret_cast<unsigned*>(lds + 2 * CHUNK_MAX + WG_WAVES * STAGE_BYTES);
            if (lane == 0) ex[wave] = nz;
            __syncthreads();
            nzw = __builtin_amdgcn_readfirstlane(ex[0] | ex[1] | ex[2] | ex[3] | 1u);
        }
        if ((HN_DBG(a) >> 8) == 1) return;   // phase timing aid
        ws.stamp(9);   // (feature generation done)
        h8 ah[16], al[16], bh[16], bl[16];   // ping-pong activation fragments
        struct Act {
            f32x16 v;
        };
        struct Frags {
            h8 hi[2], lo[2];
        };
        // Finished output fragments are parked in AGPRs until the next layer reads them: the layer that produces them
        // already holds its 128 input-fragment registers, the epilogue state and the pre-loaded side data in VGPRs, and
        // with the outputs there too the allocator goes to scratch memory inside the MFMA loops.
        auto park = [](h8& a0, h8& a1, h8& a2, h8& a3) {
#if HN_PARK_AGPR
            asm volatile("" : "+a"(a0), "+a"(a1), "+a"(a2), "+a"(a3));
#endif
        };
        auto no_pre = [](auto, const char*) { return NoData{}; };
        auto no_store = [](auto, const auto&) {};
        auto stash_frags = [&](int stash_slot) {
            return [stash_slot, &sh](auto T, const Frags& f) {
                constexpr int t = decltype(T)::value;
                sh.frag_store(stash_slot * SLOT_BYTES, 2 * t, f.hi[0], f.lo[0]);
                sh.frag_store(stash_slot * SLOT_BYTES, 2 * t + 1, f.hi[1], f.lo[1]);
            };
        };
        auto to_regs = [&](h8(&oh)[16], h8(&ol)[16]) {
            return [&oh, &ol, &park](auto T, EpiState& st, const auto&) {
                constexpr int t = decltype(T)::value;
                asm volatile("" : "+v"(st.hi[0]), "+v"(st.lo[0]), "+v"(st.hi[1]), "+v"(st.lo[1]));
                oh[2 * t] = st.hi[0];
                ol[2 * t] = st.lo[0];
                oh[2 * t + 1] = st.hi[1];
                ol[2 * t + 1] = st.lo[1];
                park(oh[2 * t], ol[2 * t], oh[2 * t + 1], ol[2 * t + 1]);
                return NoData{};
            };
        };
        // The same with the fp32 tile handed to run_layer's deferred `store` instead of being stored here: the store of tile t is
        // then issued right BEHIND the next chunk's barrier, a whole chunk ahead of the barrier after it (-DHN_DEFER_STASH=1).
        // Round-4 A/B on the C2 frame: with the a1..a7 stores REMOVED (wrong gradients, -DHN_STASH_HALF=2) the kernel
        // runs 250.5 -> 226.5 ms -- in the SAME number of cycles (in-kernel stamps) at a higher clock, 1932 -> 2052 MHz: the stores
        // cost power, not issue slots or waits; issued one chunk late (this form): no change; as half as many bytes (fp16,
        // -DHN_STASH_HALF=1, not parity-preserving): no change either.
        [[maybe_unused]] auto to_regs_hold = [&](h8(&oh)[16], h8(&ol)[16]) {
            return [&oh, &ol, &park](auto T, EpiState& st, const auto&) {
                constexpr int t = decltype(T)::value;
                asm volatile("" : "+v"(st.hi[0]), "+v"(st.lo[0]), "+v"(st.hi[1]), "+v"(st.lo[1]));
                oh[2 * t] = st.hi[0];
                ol[2 * t] = st.lo[0];
                oh[2 * t + 1] = st.hi[1];
                ol[2 * t + 1] = st.lo[1];
                park(oh[2 * t], ol[2 * t], oh[2 * t + 1], ol[2 * t + 1]);
                return Act{st.vec()};
            };
        };
        [[maybe_unused]] auto store_tile = [&](int stash_slot) {
            return [stash_slot, &sh](auto T, const Act& held) {
                if (FULL) sh.tile_store(stash_slot, decltype(T)::value, held.v);
            };
        };
        auto to_regs_keep = [&](h8(&oh)[16], h8(&ol)[16], int stash_slot) {
            return [&oh, &ol, stash_slot, &sh, &park](auto T, EpiState& st, const auto&) {
                constexpr int t = decltype(T)::value;
                asm volatile("" : "+v"(st.hi[0]), "+v"(st.lo[0]), "+v"(st.hi[1]), "+v"(st.lo[1]));
                oh[2 * t] = st.hi[0];
                ol[2 * t] = st.lo[0];
                oh[2 * t + 1] = st.hi[1];
                ol[2 * t + 1] = st.lo[1];
                park(oh[2 * t], ol[2 * t], oh[2 * t + 1], ol[2 * t + 1]);
#if defined(HN_STASH_HALF) && HN_STASH_HALF
                if (FULL) {   // (measurement builds: 1 = half-width values, see Stash::tile_store_half; 2 = no store at all: wrong gradients,
                              //  the issue-slot cost of the stores alone)
                    if constexpr (MODE == 1) {
                        if (HN_STASH_HALF == 1) sh.tile_store_half(stash_slot, t, st.vec());
                    } else
                        sh.tile_store(stash_slot, t, st.vec());
                }
#else
                if (FULL) sh.tile_store(stash_slot, t, st.vec());
#endif
                return NoData{};
            };
        };

        // ---- NB blocks of 4 output tiles over the feature space: per bone NB chunks (4 tiles x 4 k-steps each)
        //      sharing the bone's fragments, then NB leftover chunks (4 tiles x 3 k-steps [+ tail: the 4 biases,
        //      added here when BIAS]).  c1/c2[4 blk + ti] += W[tile, features] * feat.  Bone b+1's fragments are
        //      loaded while bone b's MFMAs run.
#ifndef HN_DBG_NO_FEAT_LOADS
#define HN_DBG_NO_FEAT_LOADS 0   // 1 (measurement only, WRONG results): the feature passes never read the bones' fragments back from the stash -- an
                                 // upper bound on what reading them twice instead of four times could buy (round 5, profiles/r05/README.md)
#endif
        auto bone_loads = [&](int b) { return !HN_DBG_NO_FEAT_LOADS && (b >= N_BONES || ((nz >> b) & 1u)); };   // load_bone issues its 8 loads
        auto load_bone = [&](int b, h8(&oh)[4], h8(&ol)[4]) {
            if (!HN_DBG_NO_FEAT_LOADS && (b >= N_BONES || ((nz >> b) & 1u))) {
#pragma unroll
                for (int s = 0; s < 4; ++s) sh.frag_load(feat_base, 4 * b + s, oh[s], ol[s]);
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        oh[s][jj] = (_Float16)0.f;
                        ol[s][jj] = (_Float16)0.f;
                    }
                }
            }
        };
        auto feature_pass = [&](auto NB_, auto BIAS_, auto& c1, auto& c2, auto LB_, auto NA_, auto PS_) {
            constexpr int NB = decltype(NB_)::value;
            constexpr int PS = decltype(PS_)::value;   // MFMA passes per product
            constexpr int left_bytes = decltype(LB_)::value, next_after = decltype(NA_)::value;   // chunk sizes: constants
            constexpr bool BIAS = decltype(BIAS_)::value;
            h8 fh[2][4], fl[2][4];
            // bone b's NB chunks (fragments in uh/ul); the last one prefetches the first chunk of bone nb (the next
            // bone that is run; nb == N_BONES: the leftover chunks, which follow the 21 bones in the stream)
            const int base = ws.goff - HB_BONE;   // stream offset of bone 0's first chunk (the chunk in flight)
            auto step = [&](int nb, const h8(&uh)[4], const h8(&ul)[4], h8(&nh)[4], h8(&nl)[4]) {
                static_for<NB>([&](auto BLK) {
                    constexpr int blk = decltype(BLK)::value;
                    // The next bone's 8 fragment loads are issued one step ahead, from the slot BEHIND this chunk's DMA
                    // pieces (mma_chunk's mid): they are then the youngest vector-memory operations, and the acquire of
                    // the bone's second chunk lets exactly those 8 stay in flight (vmcnt(8)) -- they get a whole chunk
                    // more to come back from HBM.  The drains are the s_waitcnt builtin (VISIBLE): with the opaque asm
                    // form the compiler guards this bone's first use with counted waits that only the loads issued
                    // AFTER it can satisfy (in-kernel stamps: blk-0 chunks 3 000 - 5 000 cycles against 2 400).
                    const bool ahead = blk == 1 && NB == 2 && HN_FEAT_LOADS_BEHIND && bone_loads(nb);
                    const char* buf = ahead ? ws.template acquire<8, HN_FEAT_ACQUIRE_VISIBLE>() : ws.template acquire<0, HN_FEAT_ACQUIRE_VISIBLE>();
                    if constexpr (blk == 0 && !(NB == 2 && HN_FEAT_LOADS_BEHIND)) load_bone(nb, nh, nl);   // nb == 21: the leftover blocks 84..86
                    if constexpr (blk + 1 < NB) {
                        ws.template begin_c<HB_BONE>();
                        if constexpr (blk == 0 && NB == 2 && HN_FEAT_LOADS_BEHIND)
                            mma_chunk<4, 4, HB_BONE, PS>(ws, buf, uh, ul, &c1[HN_EXP_ACC * blk], &c2[HN_EXP_ACC * blk], lane, [&]() { load_bone(nb, nh, nl); });
                        else
                            mma_chunk<4, 4, HB_BONE, PS>(ws, buf, uh, ul, &c1[HN_EXP_ACC * blk], &c2[HN_EXP_ACC * blk], lane);   // one pipeline over the 16 blocks
                    } else if constexpr (left_bytes <= HB_BONE) {
                        // the first leftover chunk (nb == 21) is shorter than a bone chunk: fetched at the bone size all
                        // the same (the size stays a constant; 7 - 8 KiB of the following chunk come along unused)
                        ws.goff = base + nb * (NB * HB_BONE);
                        ws.template begin_c<HB_BONE>();
                        if (nb >= N_BONES) ws.goff += left_bytes - HB_BONE;
                        mma_chunk<4, 4, HB_BONE, PS>(ws, buf, uh, ul, &c1[HN_EXP_ACC * blk], &c2[HN_EXP_ACC * blk], lane);
                    } else {
                        // (16x16x32: the leftover chunk has four k-steps and is the longer one when it carries a tail)
                        ws.goff = base + nb * (NB * HB_BONE);
                        if (nb < N_BONES) {
                            ws.template begin_c<HB_BONE>();
                            mma_chunk<4, 4, HB_BONE, PS>(ws, buf, uh, ul, &c1[HN_EXP_ACC * blk], &c2[HN_EXP_ACC * blk], lane);
                        } else {
                            ws.template begin_c<left_bytes>();
                            mma_chunk<4, 4, left_bytes, PS>(ws, buf, uh, ul, &c1[HN_EXP_ACC * blk], &c2[HN_EXP_ACC * blk], lane);
                        }
                    }
                });
            };
            load_bone(0, fh[0], fl[0]);
            unsigned rem = nzw & ~1u;
            // two steps per trip with the fragment buffers swapped, NOT one step plus a copy: with a copy at the end
            // of the trip the compiler coalesces the two buffers and sinks the next bone's loads to the point of the
            // copy, i.e. the prefetch becomes a load that the very next MFMAs wait for (measured: ~2000 cycles per bone)
            bool in_first = false;   // the leftover fragments ended up in fh[0] (else fh[1])
#pragma unroll 1
            while (true) {
                const int nb = rem ? __builtin_ctz(rem) : N_BONES;
                rem &= rem - 1u;
                step(nb, fh[0], fl[0], fh[1], fl[1]);
                if (nb == N_BONES) break;
                const int nb2 = rem ? __builtin_ctz(rem) : N_BONES;
                rem &= rem - 1u;
                step(nb2, fh[1], fl[1], fh[0], fl[0]);
                if (nb2 == N_BONES) {
                    in_first = true;
                    break;
                }
            }
            if (!in_first) {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    fh[0][s4] = fh[1][s4];
                    fl[0][s4] = fl[1][s4];
                }
            }
            static_for<NB>([&](auto BLK) {
                constexpr int blk = decltype(BLK)::value;
                const char* buf = ws.template acquire<0>();
                constexpr int nbytes = blk + 1 < NB ? left_bytes : next_after;
                ws.template begin_c<nbytes>();
                static_for<4>([&](auto TI) {
                    constexpr int ti = decltype(TI)::value;
                    if constexpr (ti == 0)
                        mma_tile<KS_LEFT, 0, nbytes, PS>(ws, buf + ti * KS_LEFT * KS_BYTES, fh[0], fl[0], c1[4 * blk + ti], c2[4 * blk + ti], lane);
                    else
                        mma_tile<KS_LEFT, 0, 0, PS>(ws, buf + ti * KS_LEFT * KS_BYTES, fh[0], fl[0], c1[4 * blk + ti], c2[4 * blk + ti], lane);
                    if constexpr (BIAS) {
                        const f32x16 bias = tail_tile(buf + 4 * KS_LEFT * KS_BYTES, ti, h);
#pragma unroll
                        for (int i = 0; i < 16; ++i) c1[4 * blk + ti][i] += bias[i];
                    }
                });
            });
        };
        // epilogue of a finished block (not overlapped with MFMAs): ph / fin as in run_layer
        auto block_epilogue = [&](auto NT_, auto& c1, auto& c2, auto&& ph, auto&& fin) {
            constexpr int NTT = decltype(NT_)::value;
            static_for<NTT>([&](auto TI) {
                constexpr int ti = decltype(TI)::value;
                EpiState st;
                arm(st);
                st.c1 = c1[ti];
                st.c2 = c2[ti];
                NoData nd;
                Epi<true, std::remove_reference_t<decltype(ph)>, NoData> epi{st, ph, nd};
                epi.run_all();
                split_finish<true>(st);
                fin(std::integral_constant<int, ti>{}, st, nd);
            });
        };
        using I2 = std::integral_constant<int, 2>;
        using I8t = std::integral_constant<int, 8>;
        using BTrue = std::integral_constant<bool, true>;
        using BFalse = std::integral_constant<bool, false>;

        float g[3] = {0.f, 0.f, 0.f}, rgb[3] = {0.f, 0.f, 0.f}, rgb1[3] = {0.f, 0.f, 0.f};
        float sdf = 0.f;
        if constexpr (!RUN_FWD) {   // the adjoint alone: the evaluation's outputs come from the forward launch
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                g[c] = a.grad[3 * nn + c];
                rgb[c] = a.rgb[3 * nn + c];
            }
        } else {
        // ---- lin0: features -> a1 (one pass, 8 tile accumulators) ---------------------------------------------
        {
            f32x16 c1[8], c2[8];
#pragma unroll
            for (int ti = 0; ti < 8; ++ti) {
                c1[ti] = zero16();
                c2[ti] = zero16();
            }
            feature_pass(I2{}, BTrue{}, c1, c2, std::integral_constant<int, HB_LEFT_T>{}, std::integral_constant<int, HB_HID>{}, IPF{});
            block_epilogue(I8t{}, c1, c2, PhSoftplus{}, to_regs_keep(ah, al, HS_A1 + 0));
        }
#if HN_DEFER_STASH
        run_layer_c<8, 16, 1, true, true, HB_HID, HB_HID, PH>(ws, ah, al, lane, h, no_pre, PhSoftplus{}, to_regs_hold(bh, bl), store_tile(HS_A1 + 1));   // lin1
#else
        run_layer_c<8, 16, 1, true, true, HB_HID, HB_HID, PH>(ws, ah, al, lane, h, no_pre, PhSoftplus{}, to_regs_keep(bh, bl, HS_A1 + 1), no_store);   // lin1
#endif
#if HN_DEFER_STASH
        run_layer_c<8, 16, 1, true, true, HB_HID, HB_HID, PH>(ws, bh, bl, lane, h, no_pre, PhSoftplus{}, to_regs_hold(ah, al), store_tile(HS_A1 + 2));   // lin2
#else
        run_layer_c<8, 16, 1, true, true, HB_HID, HB_HID, PH>(ws, bh, bl, lane, h, no_pre, PhSoftplus{}, to_regs_keep(ah, al, HS_A1 + 2), no_store);   // lin2
#endif
        // lin3 -> a4, in bh / bl like every layer's output: lin4's hidden part reads them from there.  (They used to go
        // through the stash -- 32 KB written and read back at once per tile, an HBM round trip of ~10 000 cycles in front
        // of lin4 in the in-kernel stamps.)
#if HN_DEFER_STASH
        run_layer_c<8, 16, 1, true, true, HB_HID, HB_HID, PH>(ws, ah, al, lane, h, no_pre, PhSoftplus{}, to_regs_hold(bh, bl), store_tile(HS_A1 + 3));   // lin3
#else
        run_layer_c<8, 16, 1, true, true, HB_HID, HB_HID, PH>(ws, ah, al, lane, h, no_pre, PhSoftplus{}, to_regs_keep(bh, bl, HS_A1 + 3), no_store);   // lin3
#endif
        if ((HN_DBG(a) >> 8) == 3) return;   // phase timing aid
        // ---- lin4 = [a4 | features] / sqrt2 -> a5: 8 hidden tiles (bias from the tail), then the feature pass
        {
            f32x16 c1[8], c2[8];
            static_for<8>([&](auto TI) {
                constexpr int ti = decltype(TI)::value;
                const char* buf = ws.template acquire<0>();
                constexpr int nbytes = ti < 7 ? HB_HID : HB_BONE;
                ws.template begin_c<nbytes>();
                c1[ti] = tail_tile(buf + 16 * KS_BYTES, 0, h);
                c2[ti] = zero16();
                mma_tile<16, 0, nbytes, PH>(ws, buf, bh, bl, c1[ti], c2[ti], lane);
            });
            feature_pass(I2{}, BFalse{}, c1, c2, std::integral_constant<int, HB_LEFT>{}, std::integral_constant<int, HB_HID>{}, IPF{});
            block_epilogue(I8t{}, c1, c2, PhSoftplus{}, to_regs_keep(ah, al, HS_A1 + 4));
        }
#if HN_DEFER_STASH
        run_layer_c<8, 16, 1, true, true, HB_HID, HB_HID, PH>(ws, ah, al, lane, h, no_pre, PhSoftplus{}, to_regs_hold(bh, bl), store_tile(HS_A1 + 5));   // lin5
#else
        run_layer_c<8, 16, 1, true, true, HB_HID, HB_HID, PH>(ws, ah, al, lane, h, no_pre, PhSoftplus{}, to_regs_keep(bh, bl, HS_A1 + 5), no_store);   // lin5
#endif
#if HN_DEFER_STASH
        run_layer_c<8, 16, 1, true, true, HB_HID, HB_HID, PH>(ws, bh, bl, lane, h, no_pre, PhSoftplus{}, to_regs_hold(ah, al), store_tile(HS_A1 + 6));   // lin6
#else
        run_layer_c<8, 16, 1, true, true, HB_HID, HB_HID, PH>(ws, bh, bl, lane, h, no_pre, PhSoftplus{}, to_regs_keep(ah, al, HS_A1 + 6), no_store);   // lin6
#endif
        // ---- lin7 -> a8; sdf = W8[0,:] a8 + b8; seed of the reverse sweep dz7 = sigma'(z7) W8[0,:] (scaled)
        float sdf_acc = 0.f, sdf_acc1 = 0.f;   // (16x16x32: the lane's registers of column block 0 / 1 are two samples)
        auto lin7 = [&](auto NA_) {   // NA_: the size of the chunk that follows the layer, a constant
        run_layer_c<8, 16, 1, true, true, HB_HID, decltype(NA_)::value, P7>(
            ws, ah, al, lane, h,
            [&](auto, const char* tail) { return Act{tail_tile(tail, 1, h)}; }, PhSoftplus{},
            [&](auto T, EpiState& st, const Act& w8) {
                constexpr int t = decltype(T)::value;
                asm volatile("" : "+v"(st.hi[0]), "+v"(st.lo[0]), "+v"(st.hi[1]), "+v"(st.lo[1]));
                f32x16 dz;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (S16 && ((i >> 2) & 1))
                        sdf_acc1 = fmaf(w8.v[i], st.v[i], sdf_acc1);
                    else
                        sdf_acc = fmaf(w8.v[i], st.v[i], sdf_acc);
                    dz[i] = dsoftplus_from_act(st.v[i]) * w8.v[i] * BWD_SCALE;
                }
                if (FULL) {
                    bh[2 * t] = st.hi[0];   // a8 feeds lin8
                    bl[2 * t] = st.lo[0];
                    bh[2 * t + 1] = st.hi[1];
                    bl[2 * t + 1] = st.lo[1];
                    Frags f;
                    split_tile(dz, f.hi[0], f.lo[0], f.hi[1], f.lo[1]);
                    stash_frags(HS_DZ7)(T, f);
                    if constexpr (ADJ) {
                        sh.tile_store(HS_A8, t, st.vec());
                        sh.tile_store(HS_DZ + 7, t, dz);
                    }
                }
                return NoData{};
            },
            no_store);
        };
        if constexpr (FULL) {
            lin7(std::integral_constant<int, HB_HID>{});
        } else {
            if (more)
                lin7(std::integral_constant<int, HB_BONE>{});
            else
                lin7(std::integral_constant<int, 0>{});
        }
        sdf = (S16 ? sample_sum(sdf_acc, sdf_acc1) : half_sum(sdf_acc)) + a.b8[0];
        if (!FULL) {
            if (valid && h == 0) a.sdf[n] = sdf;
            continue;
        }

        if ((HN_DBG(a) >> 8) == 5) return;   // phase timing aid
        // ---- lin8 rows 1..256: the feature vector (no activation) -> stash as fragments for colour lin0
        run_layer_c<8, 16, 1, true, true, HB_HID, HB_BWD, PH>(
            ws, bh, bl, lane, h, no_pre, PhIdentity{},
            [&](auto T, EpiState& st, const auto&) {
                constexpr int t = decltype(T)::value;
                if constexpr (S16) {
                    if (a.feat != nullptr) {
                        const int g4 = lane >> 4, n0 = tile * WG_SAMPLES + wave * 32 + (lane & 15);
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int ni = n0 + 16 * ((i >> 2) & 1);
                            if (ni < n_pts) a.feat[(size_t)ni * H + 32 * t + 16 * (i >> 3) + 4 * g4 + (i & 3)] = st.v[i];
                        }
                    }
                } else if (a.feat != nullptr && valid) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) a.feat[(size_t)n * H + 32 * t + tile_row(i, h)] = st.v[i];
                }
                stash_frags(HS_FVEC)(T, Frags{{st.hi[0], st.hi[1]}, {st.lo[0], st.lo[1]}});
                return NoData{};
            },
            no_store);

        if ((HN_DBG(a) >> 8) == 6) return;   // phase timing aid
        // ---- reverse sweep: dz_{l-1} = sigma'(z_{l-1}) * (W_l^T dz_l); sigma' from the stashed activation a_l
        auto act_of = [&](int act_slot) {
#if defined(HN_STASH_HALF) && HN_STASH_HALF
            return [&sh, act_slot](auto T, const char*) {
                if constexpr (MODE == 1)
                    return Act{sh.tile_load_half(act_slot, decltype(T)::value)};
                else
                    return Act{sh.tile_load(act_slot, decltype(T)::value)};
            };
#else
            return [&sh, act_slot](auto T, const char*) { return Act{sh.tile_load(act_slot, decltype(T)::value)}; };
#endif
        };
#pragma unroll
        for (int s = 0; s < 16; ++s) sh.frag_load(HS_DZ7 * SLOT_BYTES, s, ah[s], al[s]);
        // (adjoint mode: every dz_l also goes to the stash as an fp32 tile, slot HS_DZ + l)
        auto to_regs_t = [&](h8(&oh)[16], h8(&ol)[16], int tile_slot) {
            return [&oh, &ol, tile_slot, &sh, &park](auto T, EpiState& st, const auto&) {
                constexpr int t = decltype(T)::value;
                asm volatile("" : "+v"(st.hi[0]), "+v"(st.lo[0]), "+v"(st.hi[1]), "+v"(st.lo[1]));
                oh[2 * t] = st.hi[0];
                ol[2 * t] = st.lo[0];
                oh[2 * t + 1] = st.hi[1];
                ol[2 * t + 1] = st.lo[1];
                park(oh[2 * t], ol[2 * t], oh[2 * t + 1], ol[2 * t + 1]);
                (void)tile_slot;
                (void)sh;
                if constexpr (ADJ) sh.tile_store(tile_slot, t, st.vec());
                return NoData{};
            };
        };
        run_layer_c<8, 16, 1, false, true, HB_BWD, HB_BWD, PH>(ws, ah, al, lane, h, act_of(HS_A1 + 6), PhDsig{}, to_regs_t(bh, bl, HS_DZ + 6), no_store);   // W7^T -> dz6
        run_layer_c<8, 16, 1, false, true, HB_BWD, HB_BWD, PH>(ws, bh, bl, lane, h, act_of(HS_A1 + 5), PhDsig{}, to_regs_t(ah, al, HS_DZ + 5), no_store);   // W6^T -> dz5
        run_layer_c<8, 16, 1, false, true, HB_BWD, HB_BWD, PH>(ws, ah, al, lane, h, act_of(HS_A1 + 4), PhDsig{},                               // W5^T -> dz4 (kept)
                                         [&](auto T, EpiState& st, const auto&) {
                                             constexpr int t = decltype(T)::value;
                                             asm volatile("" : "+v"(st.hi[0]), "+v"(st.lo[0]), "+v"(st.hi[1]), "+v"(st.lo[1]));
                                             bh[2 * t] = st.hi[0];
                                             bl[2 * t] = st.lo[0];
                                             bh[2 * t + 1] = st.hi[1];
                                             bl[2 * t + 1] = st.lo[1];
                                             stash_frags(HS_DZ4)(T, Frags{{st.hi[0], st.hi[1]}, {st.lo[0], st.lo[1]}});
                                             if constexpr (ADJ) sh.tile_store(HS_DZ + 4, t, st.vec());
                                             return NoData{};
                                         },
                                         no_store);
        run_layer_c<8, 16, 1, false, true, HB_BWD, HB_BWD, PH>(ws, bh, bl, lane, h, act_of(HS_A1 + 3), PhDsig{}, to_regs_t(ah, al, HS_DZ + 3), no_store);   // W4h^T -> dz3
        run_layer_c<8, 16, 1, false, true, HB_BWD, HB_BWD, PH>(ws, ah, al, lane, h, act_of(HS_A1 + 2), PhDsig{}, to_regs_t(bh, bl, HS_DZ + 2), no_store);   // W3^T -> dz2
        run_layer_c<8, 16, 1, false, true, HB_BWD, HB_BWD, PH>(ws, bh, bl, lane, h, act_of(HS_A1 + 1), PhDsig{}, to_regs_t(ah, al, HS_DZ + 1), no_store);   // W2^T -> dz1
        run_layer_c<8, 16, 1, false, true, HB_BWD, HB_BWD, PH>(ws, ah, al, lane, h, act_of(HS_A1 + 0), PhDsig{}, to_regs_t(bh, bl, HS_DZ + 0), no_store);   // W1^T -> dz0

        if ((HN_DBG(a) >> 8) == 7) return;   // phase timing aid
        // ---- d sdf / d features contracted with the encoding Jacobian, bone by bone: first W0^T dz0 (dz0 is in
        //      bh/bl), then W4[:, 256:]^T dz4 (reloaded into ah/al)
        // G = W0^T dz0 + W4[:, 256:]^T dz4 accumulated per tile (chunks alternate between the two matrices),
        // so that the features and the Jacobian are visited once
#pragma unroll
        for (int s = 0; s < 16; ++s) sh.frag_load(HS_DZ4 * SLOT_BYTES, s, ah[s], al[s]);
        // dz0 (parked in AGPRs by the last reverse layer) is the B operand of 44 chunks: back into VGPRs once, here
#if HN_JAC_VARIANT == 0
#pragma unroll
        for (int s = 0; s < 16; ++s) asm volatile("" : "+v"(bh[s]), "+v"(bl[s]));
#elif HN_JAC_VARIANT == 2
#pragma unroll
        for (int s = 0; s < 16; ++s) asm volatile("" : "+a"(bh[s]), "+a"(bl[s]), "+a"(ah[s]), "+a"(al[s]));
#elif HN_JAC_VARIANT == 3
#pragma unroll
        for (int s = 0; s < 16; ++s) asm volatile("" : "+v"(bh[s]), "+v"(bl[s]), "+a"(ah[s]), "+a"(al[s]));
#endif
        {
            // The leftover block first (2 tiles; register 8 (u & 1) + j of tile u >> 1 <-> bone 8 u + j : (r_1 | r_2) h): its
            // per-bone values are parked in the stash and join the bone's own sums below, so that every bone is visited
            // ONCE (coordinates, d/dq, R_b^T) -- as a block behind the bones it cost a second round over all 21 of them,
            // 41 000 cycles per tile in the in-kernel stamps.
            {
                f32x16 L1[2], L2[2];
                static_for<2>([&](auto U) {
                    constexpr int u = decltype(U)::value;
                    const char* buf0 = ws.template acquire<0>();
                    ws.template begin_c<HB_BWD>();
                    L1[u] = zero16();
                    L2[u] = zero16();
                    mma_tile<16, 0, HB_BWD, PJ>(ws, buf0, bh, bl, L1[u], L2[u], lane);
                    const char* buf4 = ws.template acquire<0>();
                    ws.template begin_c<HB_BWD>();   // after the last one: bone 0's first chunk
                    mma_tile<16, 0, HB_BWD, PJ>(ws, buf4, ah, al, L1[u], L2[u], lane);
                });
                f32x16 La = combine(L1[0], L2[0]), Lb = combine(L1[1], L2[1]);
                tile_out(La);
                tile_out(Lb);
                static_for<N_BONES>([&](auto B_) {
                    constexpr int b = decltype(B_)::value;
                    if ((nz >> b) & 1u) sh.f32_store(LEFTJ + b * 256, b < 16 ? La[b] : Lb[b - 16]);
                });
            }
            const int jbase = ws.goff - HB_BWD;   // stream offset of bone 0's first chunk (in flight)
            unsigned rem = nzw & ~1u;
            int b = 0;
#pragma unroll 1
            while (b < N_BONES) {
                const int nb = rem ? __builtin_ctz(rem) : N_BONES;   // the next bone that is run (21: what follows the bones: colour lin0's first chunk, the same size)
                rem &= rem - 1u;
                f32x16 G1[2], G2[2];
                const bool live = (nz >> b) & 1u;
                // the bone's row of the leftover block: requested four chunks ahead of its use
                const float Gl = live ? sh.f32_load(LEFTJ + b * 256) : 0.f;
                static_for<2>([&](auto U) {
                    constexpr int u = decltype(U)::value;
                    const char* buf0 = ws.template acquire<0>();
                    ws.template begin_c<HB_BWD>();
                    // the bone's own features go stash -> LDS by DMA (no registers: loaded into VGPRs here they
                    // are spilled one load at a time, 8 serialised round trips per bone); they land under the
                    // bone's MFMAs, the next acquire's vmcnt(0) covers them
                    if (u == 0 && live) {
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(sh.rsrc, (lds_void_t*)(stage + i * 1024), 16, lane_x16(),
                                                                     FEAT + (4 * b + (i >> 1)) * KS_BYTES + (i & 1) * 1024, 0, STASH_AUX);
                    }
                    G1[u] = zero16();
                    G2[u] = zero16();
                    mma_tile<16, 0, HB_BWD, PJ>(ws, buf0, bh, bl, G1[u], G2[u], lane);
                    const char* buf4 = ws.template acquire<0>();
                    if constexpr (u == 1) ws.goff = jbase + nb * (4 * HB_BWD);
                    ws.template begin_c<HB_BWD>();
                    mma_tile<16, 0, HB_BWD, PJ>(ws, buf4, ah, al, G1[u], G2[u], lane);
                });
                if (live) {   // a bone whose mask is 0 for the whole wave contributes exactly 0
                    const Bone2 bn = coords(b);
                    const float kk = -TAU2 * (1.f - bn.hh);
                    float own[4][8];
                    f32x16 Ga = combine(G1[0], G2[0]), Gb = combine(G1[1], G2[1]);
                    tile_out(Ga);   // (16x16x32: back to the lane <-> sample map of the contraction below)
                    tile_out(Gb);
                    {
                        h8 fh[4], fl[4];
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            fh[s] = *reinterpret_cast<const h8*>(stage + (2 * s) * 1024 + lane * 16);
                            fl[s] = *reinterpret_cast<const h8*>(stage + (2 * s + 1) * 1024 + lane * 16);
                        }
#pragma unroll
                        for (int s = 0; s < 4; s += 2) {
                            frags_out(fh[s], fh[s + 1]);
                            frags_out(fl[s], fl[s + 1]);
                        }
#pragma unroll
                        for (int s = 0; s < 4; ++s)
#pragma unroll
                            for (int jj = 0; jj < 8; ++jj) own[s][jj] = unsplit(fh[s][jj], fl[s][jj]);
                    }
                    // the leftover pair (r_1 | r_2) h of this bone: own value and its row of the leftover block (Gl)
                    const float ownl = (h ? bn.r[2] : bn.r[1]) * bn.hh;
                    if constexpr (ADJ) {
                        // the same contraction through the h-weighted sums, which the adjoint's second-order term needs
                        // again (unscaled: G carries BWD_SCALE)
                        BoneSums S;
                        bone_sums<true>(Ga, Gb, own, bn.hh, h, S);
                        S.T0 = fmaf(Gl, ownl, S.T0);
                        S.T1[2] += h ? 0.f : Gl * bn.hh;
                        S.T1[3] += h ? Gl * bn.hh : 0.f;
                        float Sv = fmaf(kk, S.T0, S.T1[0]);
                        float Sr[3] = {S.T1[1], S.T1[2], S.T1[3]};
                        to_p(Sv, Sr, bn, b, g);
                        sums_reduce(S, true);
                        sh.f32_store(TS + (9 * b) * 256, S.T0 * BWD_INV);
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) {
                            sh.f32_store(TS + (9 * b + 1 + q4) * 256, S.T1[q4] * BWD_INV);
                            sh.f32_store(TS + (9 * b + 5 + q4) * 256, S.T2[q4] * BWD_INV);
                        }
                    } else {
                        float Sv = 0.f, Sr[3] = {0.f, 0.f, 0.f};
                        bone_jacobian(Ga, Gb, own, bn, kk, h, Sv, Sr);
                        Sv = fmaf(Gl * ownl, kk, Sv);
                        Sr[1] += h ? 0.f : Gl * bn.hh;
                        Sr[2] += h ? Gl * bn.hh : 0.f;
                        to_p(Sv, Sr, bn, b, g);
                    }
                }
                b = nb;
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) g[c] *= BWD_INV;

        if ((HN_DBG(a) >> 8) == 8) return;   // phase timing aid
        // ---- colour lin0 = [features | feature vector | enc(g)] -> relu: per pass 4 feature-vector tiles, the
        //      feature block, then the enc(g) chunk (whose tail holds the 4 biases)
        h8 gh[2], gl[2];
        {
            float fg[2][8];
            encode_v4h(g, h, fg);
            split8(fg[0], gh[0], gl[0]);
            split8(fg[1], gh[1], gl[1]);
            frags_in(gh[0], gh[1]);
            frags_in(gl[0], gl[1]);
        }
        {
            f32x16 c1[8], c2[8];
            {
                h8 xh[16], xl[16];
#pragma unroll
                for (int s = 0; s < 16; ++s) sh.frag_load(HS_FVEC * SLOT_BYTES, s, xh[s], xl[s]);
                static_for<8>([&](auto TI) {
                    constexpr int ti = decltype(TI)::value;
                    const char* buf = ws.template acquire<0>();
                    constexpr int nbytes = ti < 7 ? HB_BWD : HB_BONE;
                    ws.template begin_c<nbytes>();
                    c1[ti] = zero16();
                    c2[ti] = zero16();
                    mma_tile<16, 0, nbytes, PC>(ws, buf, xh, xl, c1[ti], c2[ti], lane);
                });
            }
            feature_pass(I2{}, BFalse{}, c1, c2, std::integral_constant<int, HB_LEFT>{}, std::integral_constant<int, HB_G>{}, IPC{});
            static_for<2>([&](auto BLK) {
                constexpr int blk = decltype(BLK)::value;
                const char* buf = ws.template acquire<0>();
                constexpr int nbytes = blk == 0 ? HB_G : HB_HID;
                ws.template begin_c<nbytes>();
                ws.template pieces_all_c<nbytes>();   // 6 MFMA slots per tile here: too few to spread the pieces over
                const char* tail = buf + 4 * 2 * KS_BYTES;
                static_for<4>([&](auto TI) {
                    constexpr int ti = decltype(TI)::value;
                    mma_tile<2, 0, 0>(ws, buf + ti * 2 * KS_BYTES, gh, gl, c1[4 * blk + ti], c2[4 * blk + ti], lane);
                    const f32x16 bias = tail_tile(tail, ti, h);
#pragma unroll
                    for (int i = 0; i < 16; ++i) c1[4 * blk + ti][i] += bias[i];
                });
            });
            block_epilogue(I8t{}, c1, c2, PhRelu{}, to_regs_t(bh, bl, HS_C + 0));
        }
        run_layer_c<8, 16, 1, true, true, HB_HID, HB_HID, PC>(ws, bh, bl, lane, h, no_pre, PhRelu{}, to_regs_t(ah, al, HS_C + 1), no_store);   // colour lin1
        run_layer_c<8, 16, 1, true, true, HB_HID, HB_HID, PC>(ws, ah, al, lane, h, no_pre, PhRelu{}, to_regs_t(bh, bl, HS_C + 2), no_store);   // colour lin2
        struct W3 {
            f32x16 w[3];
        };
        auto col3 = [&](auto NA_) {
        run_layer_c<8, 16, 1, true, false, HB_HID, decltype(NA_)::value, PC>(   // colour lin3 + the 3 rows of lin4 (tail slots 1..3)
            ws, bh, bl, lane, h,
            [&](auto, const char* tail) { return W3{{tail_tile(tail, 1, h), tail_tile(tail, 2, h), tail_tile(tail, 3, h)}}; },
            PhRelu{},
            [&](auto T, EpiState& st, const W3& w) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        if (S16 && ((i >> 2) & 1))
                            rgb1[c] = fmaf(w.w[c][i], st.v[i], rgb1[c]);
                        else
                            rgb[c] = fmaf(w.w[c][i], st.v[i], rgb[c]);
                    }
                asm volatile("" : "+v"(rgb[0]), "+v"(rgb[1]), "+v"(rgb[2]));
                if constexpr (S16) asm volatile("" : "+v"(rgb1[0]), "+v"(rgb1[1]), "+v"(rgb1[2]));
                if constexpr (ADJ) sh.tile_store(HS_C + 3, decltype(T)::value, st.vec());
                return NoData{};
            },
            no_store);
        };
        if constexpr (MODE == 2) {
            col3(std::integral_constant<int, HB_W4ROWS>{});
        } else {
            if (more)
                col3(std::integral_constant<int, HB_BONE>{});
            else
                col3(std::integral_constant<int, 0>{});
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[c] = sigmoid_fast((S16 ? sample_sum(rgb[c], rgb1[c]) : half_sum(rgb[c])) + a.c_blast[c]);
        }   // RUN_FWD
        if constexpr (RUN_ADJ) {
#include "hn_field2_hand_adj.inl"
            if (a.pose_part != nullptr) {   // this tile's two rows of this wave: [first frame | the other frame]
                __builtin_amdgcn_wave_barrier();
                float* row = a.pose_part + ((size_t)tile * WG_WAVES + wave) * POSE_FRAMES * (N_BONES * 12);
                for (int f = 0; f < POSE_FRAMES; ++f)
                    for (int i = lane; i < N_BONES * 12; i += 64) row[f * (N_BONES * 12) + i] = prow[f * POSE_ROW + i];
                __builtin_amdgcn_wave_barrier();
            }
            continue;
        }
        if (valid && h == 0) {
            a.sdf[n] = sdf;
            a.grad[3 * n] = g[0];
            a.grad[3 * n + 1] = g[1];
            a.grad[3 * n + 2] = g[2];
            a.rgb[3 * n] = rgb[0];
            a.rgb[3 * n + 1] = rgb[1];
            a.rgb[3 * n + 2] = rgb[2];
        }
    }
}

// out += the rows of part [n_tiles * 4][2][21 * 12] that belong to frame `blockIdx.x`, added in a FIXED order: g_bt_inv [21,4,4] rows
// 0..2 (12 values per bone), g_T_pose [21,3] = minus the translation column's sums (the statement of the adjoint kernel's atomics,
// hn_field2_hand_adj.inl).  The frame's rows are those of the waves whose FIRST sample is the frame's (slot 0) -- waves w_a .. w_b - 1,
// consecutive because the lists keep the dense order -- plus slot 1 of wave w_a - 1 where that wave crossed into the frame (a dense
// list whose frames are not wave-aligned).  1024 threads = 252 outputs x 4 row quarters of [w_a, w_b); a thread adds its quarter's rows
// into 8 interleaved accumulators (row w_a + r into accumulator r mod 8: eight loads in flight instead of one dependent chain), folds
// them 0..7, the quarters are folded 0..3 through LDS, the crossing wave's share is added last.  Every index is RELATIVE to the
// frame's first wave and the count is the frame's own: a frame's sums are the same bits whether it is alone in the launch or one of
// several (frame-aligned compact lists, and dense lists with pts_per_frame a multiple of 128: fitting.BatchedSingleFit).
//   dense list:            w_a = ceil(f ppf / 32), w_b = ceil((f + 1) ppf / 32)
//   frame-aligned compact: w_a = seg[f] / 32, w_b = w_a + 4 ceil(live_f / 128)       (a tile that holds only the stand-in adds zeros)
//   plain compact (1 frame): w_a = 0, w_b = 4 ceil((n_dev - 1) / 128)
static __global__ __launch_bounds__(1024) void k_pose_part_reduce(const float* __restrict__ part, int n_pts, int pts_per_frame,
                                                                  const int* __restrict__ n_pts_dev, const int* __restrict__ seg,
                                                                  float* __restrict__ g_bt_inv, float* __restrict__ g_T_pose) {
    constexpr int W = N_BONES * 12;
    const int n_frames = gridDim.x, fr = blockIdx.x;
    if (g_bt_inv != nullptr) g_bt_inv += (size_t)fr * N_BONES * 16;
    if (g_T_pose != nullptr) g_T_pose += (size_t)fr * N_BONES * 3;
    __shared__ float q[4][W];
    int wa, wb;
    bool crossing = false;
    if (n_pts_dev == nullptr) {
        const long long lo = (long long)fr * pts_per_frame, hi = lo + pts_per_frame < n_pts ? lo + pts_per_frame : n_pts;
        wa = (int)((lo + 31) / 32);
        wb = (int)((hi + 31) / 32);
        crossing = (lo % 32) != 0;   // wave wa - 1 starts in frame fr - 1 and ends in this one
    } else if (seg != nullptr) {
        wa = seg[fr] / 32;
        wb = wa + ((seg[n_frames + 1 + fr] + WG_SAMPLES - 1) / WG_SAMPLES) * WG_WAVES;
    } else {
        wa = 0;
        wb = ((n_pts_dev[0] - 1 + WG_SAMPLES - 1) / WG_SAMPLES) * WG_WAVES;
    }
    const size_t pitch = (size_t)POSE_FRAMES * W;
    const int rows = wb - wa;
    const int t = threadIdx.x % 256, qi = threadIdx.x / 256;
    if (t < W) {
        const float* p0 = part + (size_t)wa * pitch + t;
        const int per = (rows + 3) / 4, r0 = qi * per, r1 = r0 + per < rows ? r0 + per : rows;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int r = r0;
        for (; r + 8 <= r1; r += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += p0[(size_t)(r + u) * pitch];
        }
        for (int u = 0; r < r1; ++r, ++u) acc[u] += p0[(size_t)r * pitch];
        float sum = acc[0];
#pragma unroll
        for (int u = 1; u < 8; ++u) sum += acc[u];
        q[qi][t] = sum;
    }
    __syncthreads();
    if (qi == 0 && t < W) {
        float acc = ((q[0][t] + q[1][t]) + q[2][t]) + q[3][t];
        if (crossing && wa > 0) acc += part[(size_t)(wa - 1) * pitch + W + t];
        const int b = t / 12, k = t % 12;
        if (g_bt_inv != nullptr) g_bt_inv[b * 16 + k] += acc;
        if (g_T_pose != nullptr && (k & 3) == 3) g_T_pose[b * 3 + (k >> 2)] -= acc;
    }
}

// the kernels proper: k_field2_hand<MODE> (fp32-equivalent, the names the profiles of every round carry) and the
// single-pass evaluation kernels of HN_PREC_F16
template <int MODE>
__global__ __launch_bounds__(256) void k_field2_hand(const Hand2Args a) {
    field2_hand_body<MODE, 3>(a);
}
template <int MODE>
__global__ __launch_bounds__(256) void k_field2_hand_f16(const Hand2Args a) {
    field2_hand_body<MODE, 1>(a);
}

constexpr size_t HAND2_LDS = HAND2_LDS_POSE + WG_WAVES * POSE_FRAMES * POSE_ROW * sizeof(float);   // + the waves' pose-gradient rows (adjoint modes)

static int hand2_grid(int n_pts, int n_cus) {
    const int n_tiles = (n_pts + WG_SAMPLES - 1) / WG_SAMPLES;
    return n_tiles < n_cus ? n_tiles : n_cus;
}

// Grid of the TAPED evaluation and of the adjoint from a tape (modes 3 / 4: their stash is indexed by tile, so any grid works).  These
// two run in PAIRS -- the hand's beside the object's, on two streams -- and a persistent grid of one workgroup per CU never gives a CU
// back before its last tile: with more tiles than CUs in the first kernel (several frames side by side, fitting.fit_frames_batched)
// the second kernel's small launches and then its tiles waited for the first one's tail (measured, 4 frames: k_sample_points_t 1.25 ms
// in front of k_field2_obj<3>, the phase 3.45 ms for 2.7 ms of tile time).  One workgroup per TILE instead: the dispatcher hands
// freed CUs to whichever kernel has workgroups waiting, and the tail is one tile.  HN_TAPED_GRID=cus: the persistent grid (A/B).
static int taped_grid(int n_pts, int n_cus) {
    static const int mode = [] {
        const char* e = getenv("HN_TAPED_GRID");
        return (e != nullptr && e[0] == 'c') ? 0 : 1;
    }();
    const int n_tiles = (n_pts + WG_SAMPLES - 1) / WG_SAMPLES;
    if (mode == 0) return n_tiles < n_cus ? n_tiles : n_cus;
    return n_tiles < 65535 ? n_tiles : 65535;
}

static void hand2_common_args(Hand2Args& a, const hn_field* f, const float* pts, int n_pts, const float* bt_inv, const float* T_pose,
                              int n_frames, int pts_per_frame, void* workspace) {
    a.pts = pts;
    a.bt_inv = bt_inv;
    a.T_pose = T_pose;
    a.n_pts = n_pts;
    a.pts_per_frame = pts_per_frame;
    a.n_frames = n_frames;
    a.b8 = f->raw_sdf_b[8];
    a.c_blast = f->raw_col_b[4];
    a.scratch = reinterpret_cast<float4*>(workspace);
    a.dbg = 0;
    a.cull = f->cull_far_field;
    a.xsync = nullptr;
    a.n_pts_dev = launch_n_pts_dev();   // (set by the caller around a launch over a compacted sample list, else NULL)
    a.orig_idx = launch_orig_idx();
}

#if defined(HN_HAND_QUAD_TU)    // hn_field2_hand_q.hip: only the device helpers above are wanted
#elif defined(HN_HAND_F16_TU)   // hn_field2_hand_f16.hip: the evaluation kernels of HN_PREC_F16 (single-pass hidden layers)
int launch_field2_hand_f16(const Hand2Args& a, int grid, bool full, hipStream_t stream) {
    static std::atomic<uint64_t> lds_full{0}, lds_sdf{0};
    HN_TRY_RC(ensure_dynamic_lds(reinterpret_cast<const void*>(k_field2_hand_f16<1>), (int)HAND2_LDS, &lds_full));
    HN_TRY_RC(ensure_dynamic_lds(reinterpret_cast<const void*>(k_field2_hand_f16<0>), (int)HAND2_LDS, &lds_sdf));
    if (full)
        hipLaunchKernelGGL(k_field2_hand_f16<1>, dim3(grid), dim3(256), HAND2_LDS, stream, a);
    else
        hipLaunchKernelGGL(k_field2_hand_f16<0>, dim3(grid), dim3(256), HAND2_LDS, stream, a);
    HN_LAUNCH_CHECK();
    return HN_OK;
}
#elif !defined(HN_HAND_ADJ_TU)   // this translation unit: the evaluation kernels (MODE 0, 1); hn_field2_hand_adj.hip: MODE 2
int launch_field2_hand_f16(const Hand2Args& a, int grid, bool full, hipStream_t stream);
int launch_field2_hand_q(const Hand2Args& a, int n_blocks, int n_cus, hipStream_t stream);   // hn_field2_hand_q.hip
size_t field2_hand_q_workspace_bytes(int n_blocks, int n_cus);
// sdf-only launches of at most this many 32-sample blocks take the latency form (four waves per block): up to two rounds of
// quarter-length blocks beat one round of whole tiles.  hn_debug_quad_max_blocks(0) switches it off (tests, A/B runs).
static int quad_max_blocks(int n_cus) {
    const int v = quad_max_blocks_override();   // hn_debug_quad_max_blocks; -1: the default
    return v >= 0 ? v : 2 * n_cus;
}
size_t field2_hand_workspace_bytes(int n_pts, int n_cus) {
    return (size_t)hand2_grid(n_pts, n_cus) * WG_WAVES * HAND2_SLOTS * SLOT_F4 * sizeof(float4) + 256;   // + the XCD pacing counters
}

int launch_field2_hand_taped(const hn_field* f, const float* pts, int n_pts, const float* bt_inv, const float* T_pose, int n_frames,
                             int pts_per_frame, float* sdf, float* grad, float* rgb, float* feat, void* tape, size_t tape_bytes, hipStream_t stream);

// tape != NULL (full evaluation only): the evaluation keeps its tape there (field2_hand_tape_bytes) for a later
// adjoint launch, instead of using the workspace (k_field2_hand<3>, compiled in hn_field2_hand_adj.hip)
int launch_field2_hand(const hn_field* f, const float* pts, int n_pts, const float* bt_inv, const float* T_pose,
                       int n_frames, int pts_per_frame, float* sdf, float* grad, float* rgb, float* feat, void* workspace,
                       size_t workspace_bytes, bool full, hipStream_t stream, void* tape = nullptr, size_t tape_bytes = 0) {
    if (full && tape != nullptr && n_pts > 0)
        return launch_field2_hand_taped(f, pts, n_pts, bt_inv, T_pose, n_frames, pts_per_frame, sdf, grad, rgb, feat, tape, tape_bytes, stream);
    if (n_pts <= 0) return HN_OK;
    HN_REQUIRE(bt_inv != nullptr && T_pose != nullptr && n_frames >= 1 && pts_per_frame >= 1,
               "hand field needs bt_inv / T_pose and frame sizes");
    Hand2Args a{};
    hand2_common_args(a, f, pts, n_pts, bt_inv, T_pose, n_frames, pts_per_frame, workspace);
    a.blob = reinterpret_cast<const char*>(full ? f->v2_full : f->v2_sdf);
    a.blob_bytes = full ? f->v2_full_bytes : f->v2_sdf_bytes;
    if (a.blob == nullptr) {
        set_error("field was not created with HN_PREC_F16X3");
        return HN_EINVAL;
    }
    a.sdf = sdf;
    a.grad = grad;
    a.rgb = rgb;
    a.feat = feat;
#ifdef HN_DEBUG_HOOKS
    {
        const char* e = getenv("HN_DBG");
        a.dbg = e ? atoi(e) : 0;
    }
#endif
    int n_cus = device_cus();
    if (n_cus <= 0) n_cus = 256;
    const int grid = hand2_grid(n_pts, n_cus);
    const size_t need = (size_t)grid * WG_WAVES * HAND2_SLOTS * SLOT_F4 * sizeof(float4);
    if (workspace == nullptr || workspace_bytes < need) {
        set_error("field workspace too small: %zu < %zu", workspace_bytes, need);
        return HN_ENOMEM;
    }
    // a launch too small to fill the chip: the latency form (bit-identical results; its stash fits the same workspace)
    const int n_blocks = (n_pts + 31) / 32;
    if (!full && !f->single_pass && n_blocks <= quad_max_blocks(n_cus) && field2_hand_q_workspace_bytes(n_blocks, n_cus) <= workspace_bytes)
        return launch_field2_hand_q(a, n_blocks, n_cus, stream);
    // XCD pacing: launches of many tiles per workgroup (the image-sized ones), where the workspace has the room
    if (HN_XCD_PACING && (n_pts + WG_SAMPLES - 1) / WG_SAMPLES >= XCD_PACE_MIN_ROUNDS * grid && workspace_bytes >= need + 64) {
        a.xsync = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(workspace) + need);
        HN_CHECK_HIP(hipMemsetAsync(a.xsync, 0, 64, stream));
        if (const int ph = pace_phantom_members()) HN_CHECK_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(a.xsync), ph, 8, stream));
    }
    if (f->single_pass) return launch_field2_hand_f16(a, grid, full, stream);   // HN_PREC_F16
    static std::atomic<uint64_t> lds_full{0}, lds_sdf{0};   // devices on which the LDS size attribute is set
    HN_TRY_RC(ensure_dynamic_lds(reinterpret_cast<const void*>(k_field2_hand<1>), (int)HAND2_LDS, &lds_full));
    HN_TRY_RC(ensure_dynamic_lds(reinterpret_cast<const void*>(k_field2_hand<0>), (int)HAND2_LDS, &lds_sdf));
    if (full)
        hipLaunchKernelGGL(k_field2_hand<1>, dim3(grid), dim3(256), HAND2_LDS, stream, a);
    else
        hipLaunchKernelGGL(k_field2_hand<0>, dim3(grid), dim3(256), HAND2_LDS, stream, a);
    HN_LAUNCH_CHECK();
    return HN_OK;
}
#else
static size_t pose_part_bytes(int n_pts) {   // the rows of Hand2Args::pose_part (0: too many tiles, atomics)
    const size_t n_tiles = ((size_t)(n_pts > 0 ? n_pts : 0) + WG_SAMPLES - 1) / WG_SAMPLES;
    if (n_tiles > (size_t)POSE_MAX_TILES) return 0;
    return ((n_tiles * WG_WAVES * POSE_FRAMES * N_BONES * 12 * sizeof(float)) + 255) & ~size_t(255);
}
int field2_hand_signal_arrays() { return HSG_COUNT; }
int hand_dropped_samples(unsigned long long* n, bool reset) {   // (waits for the device: diagnostics, not a launch path)
    unsigned long long v = 0ull;
    HN_CHECK_HIP(hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_hn_dropped_samples), sizeof(v)));
    if (reset) {
        const unsigned long long z = 0ull;
        HN_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_hn_dropped_samples), &z, sizeof(z)));
    }
    if (n != nullptr) *n = v;
    return HN_OK;
}
size_t field2_hand_pose_rows_bytes(int n_pts) { return pose_part_bytes(n_pts); }
size_t field2_hand_adj_workspace_bytes(int n_pts, int n_cus) {
    const int grid = hand2_grid(n_pts, n_cus);
    return (size_t)grid * WG_WAVES * HAND2_SLOTS_ADJ * SLOT_F4 * sizeof(float4) + pose_part_bytes(n_pts);
}

// bytes of the tape a taped full evaluation leaves for the adjoint launch: the adjoint's stash slots per sample TILE
size_t field2_hand_tape_bytes(int n_pts) {
    const size_t n_tiles = ((size_t)(n_pts > 0 ? n_pts : 0) + WG_SAMPLES - 1) / WG_SAMPLES;
    return n_tiles * WG_WAVES * HAND2_SLOTS_ADJ * SLOT_F4 * sizeof(float4);
}

// the full evaluation that keeps its tape (MODE 3)
int launch_field2_hand_taped(const hn_field* f, const float* pts, int n_pts, const float* bt_inv, const float* T_pose, int n_frames,
                             int pts_per_frame, float* sdf, float* grad, float* rgb, float* feat, void* tape, size_t tape_bytes, hipStream_t stream) {
    HN_REQUIRE(bt_inv != nullptr && T_pose != nullptr && n_frames >= 1 && pts_per_frame >= 1,
               "hand field needs bt_inv / T_pose and frame sizes");
    HN_REQUIRE(f->v2_full != nullptr, "field was not created with HN_PREC_F16X3");
    if (tape_bytes < field2_hand_tape_bytes(n_pts)) {
        set_error("field tape too small: %zu < %zu", tape_bytes, field2_hand_tape_bytes(n_pts));
        return HN_ENOMEM;
    }
    Hand2Args a{};
    hand2_common_args(a, f, pts, n_pts, bt_inv, T_pose, n_frames, pts_per_frame, tape);
    a.blob = reinterpret_cast<const char*>(f->v2_tape != nullptr ? f->v2_tape : f->v2_full);
    a.blob_bytes = f->v2_tape != nullptr ? f->v2_tape_bytes : f->v2_full_bytes;
    a.sdf = sdf;
    a.grad = grad;
    a.rgb = rgb;
    a.feat = feat;   // (NULL, or the feature vector [n, 256]: the parameter-gradient path pairs it with the colour network's first layer)
    int n_cus = device_cus();
    if (n_cus <= 0) n_cus = 256;
    static std::atomic<uint64_t> lds_tape{0};
    HN_TRY_RC(ensure_dynamic_lds(reinterpret_cast<const void*>(k_field2_hand<3>), (int)HAND2_LDS, &lds_tape));
    hipLaunchKernelGGL(k_field2_hand<3>, dim3(taped_grid(n_pts, n_cus)), dim3(256), HAND2_LDS, stream, a);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

// hn_field_eval_bwd for an HN_PREC_F16X3 hand field: one persistent launch (evaluation + adjoint per sample tile).
// g_bt_inv / g_T_pose are accumulated into (the caller zeroes them).  tape != NULL: the adjoint alone (MODE 4) from the
// tape of a taped evaluation of the same points, whose outputs `grad`, `rgb` are passed back in.
int launch_field2_hand_adj(const hn_field* f, const float* pts, int n_pts, const float* bt_inv, const float* T_pose, int n_frames,
                           int pts_per_frame, const float* g_sdf, const float* g_grad, const float* g_rgb, float* g_pts,
                           float* g_bt_inv, float* g_T_pose, void* workspace, size_t workspace_bytes, hipStream_t stream,
                           const void* tape = nullptr, const float* grad = nullptr, const float* rgb = nullptr, float* sig = nullptr,
                           size_t sig_pitch = 0, float* gb_out = nullptr) {
    if (n_pts <= 0) return HN_OK;
    HN_REQUIRE((tape != nullptr ? f->v2_adjonly : f->v2_adj) != nullptr, "field has no adjoint program");
    HN_REQUIRE(tape == nullptr || (grad != nullptr && rgb != nullptr), "the adjoint from a tape needs the evaluation's grad / rgb");
    HN_REQUIRE(sig == nullptr || (tape != nullptr && gb_out != nullptr && sig_pitch >= (size_t)n_pts * 256), "the signal arrays belong to the adjoint from a tape");
    Hand2Args a{};
    a.sig = sig;
    a.sig_pitch = sig_pitch;
    a.gb_out = gb_out;
    hand2_common_args(a, f, pts, n_pts, bt_inv, T_pose, n_frames, pts_per_frame, workspace);
    a.blob = reinterpret_cast<const char*>(f->v2_adj);
    a.blob_bytes = f->v2_adj_bytes;
    a.g_sdf = g_sdf;
    a.g_grad = g_grad;
    a.g_rgb = g_rgb;
    a.g_pts = g_pts;
    a.g_bt_inv = g_bt_inv;
    a.g_T_pose = g_T_pose;
    int n_cus = device_cus();
    if (n_cus <= 0) n_cus = 256;
    const int grid = hand2_grid(n_pts, n_cus);
    // Every wave sums the pose-gradient addends of each of its tiles in LDS -- per frame: its first sample's and, where a wave of a
    // dense multi-frame list crosses a frame boundary, the next one's -- and writes the two rows per tile; the rows of a frame are
    // added in a fixed order behind the launch (k_pose_part_reduce): bit-reproducible pose gradients that do not depend on the
    // launch's other frames, and no atomics in the kernel (they cost it 0.17 ms of 0.93: every wave's 15 adds per bone on the same
    // 315 addresses stood in front of the next chunk's vmcnt(0)).  Any number of frames of at least 32 samples (a wave then touches
    // at most two), a compact multi-frame list only in its frame-aligned layout (hn_api.hip), up to POSE_MAX_TILES tiles; otherwise
    // float atomics.  The rows live behind the stash in the workspace (the adjoint from a tape has the whole workspace free).
    const size_t stash_bytes = (size_t)grid * WG_WAVES * HAND2_SLOTS_ADJ * SLOT_F4 * sizeof(float4);
    a.frame_seg = launch_frame_seg();
    const size_t rows_bytes = pose_part_bytes(n_pts);
    const bool frames_ok = n_frames == 1 || (a.n_pts_dev == nullptr ? pts_per_frame >= 32 : a.frame_seg != nullptr);
    const bool det = n_frames >= 1 && frames_ok && rows_bytes != 0 && (g_bt_inv != nullptr || g_T_pose != nullptr) && workspace != nullptr &&
                     workspace_bytes >= (tape != nullptr ? rows_bytes : stash_bytes + rows_bytes);
    if (det) a.pose_part = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + (tape != nullptr ? 0 : stash_bytes));
    auto reduce_rows = [&]() {
        if (!det) return;
        hipLaunchKernelGGL(k_pose_part_reduce, dim3(n_frames), dim3(1024), 0, stream, a.pose_part, n_pts, pts_per_frame, a.n_pts_dev, a.frame_seg, g_bt_inv,
                           g_T_pose);
    };
    if (tape != nullptr) {
        a.blob = reinterpret_cast<const char*>(f->v2_adjonly);
        a.blob_bytes = f->v2_adjonly_bytes;
        a.scratch = reinterpret_cast<float4*>(const_cast<void*>(tape));
        a.grad = const_cast<float*>(grad);   // read only in this mode
        a.rgb = const_cast<float*>(rgb);
        if (sig != nullptr) {   // MODE 5: the same adjoint, leaving the per-layer signals of the parameter gradients
            static std::atomic<uint64_t> lds_sig{0};
            HN_TRY_RC(ensure_dynamic_lds(reinterpret_cast<const void*>(k_field2_hand<5>), (int)HAND2_LDS, &lds_sig));
            hipLaunchKernelGGL(k_field2_hand<5>, dim3(taped_grid(n_pts, n_cus)), dim3(256), HAND2_LDS, stream, a);
            reduce_rows();
            HN_LAUNCH_CHECK();
            return HN_OK;
        }
        static std::atomic<uint64_t> lds_adjonly{0};
        HN_TRY_RC(ensure_dynamic_lds(reinterpret_cast<const void*>(k_field2_hand<4>), (int)HAND2_LDS, &lds_adjonly));
        hipLaunchKernelGGL(k_field2_hand<4>, dim3(taped_grid(n_pts, n_cus)), dim3(256), HAND2_LDS, stream, a);
        reduce_rows();
        HN_LAUNCH_CHECK();
        return HN_OK;
    }
    const size_t need = stash_bytes;
    if (workspace == nullptr || workspace_bytes < need) {
        set_error("adjoint workspace too small: %zu < %zu", workspace_bytes, need);
        return HN_ENOMEM;
    }
    static std::atomic<uint64_t> lds_adj{0};
    HN_TRY_RC(ensure_dynamic_lds(reinterpret_cast<const void*>(k_field2_hand<2>), (int)HAND2_LDS, &lds_adj));
    hipLaunchKernelGGL(k_field2_hand<2>, dim3(grid), dim3(256), HAND2_LDS, stream, a);
    reduce_rows();
    HN_LAUNCH_CHECK();
    return HN_OK;
}
#endif

}  // namespace v2
}  // namespace hn

#if defined(HN_TS) && defined(HN_HAND_ADJ_TU)
extern "C" int hn_debug_ts_adj(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(hn::v2::g_hn_ts), sizeof(unsigned long long) * n);
}
#endif
#if defined(HN_TS) && !defined(HN_HAND_ADJ_TU) && !defined(HN_HAND_QUAD_TU) && !defined(HN_HAND_F16_TU)
extern "C" int hn_debug_ts(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(hn::v2::g_hn_ts), sizeof(unsigned long long) * n);
}
#endif
