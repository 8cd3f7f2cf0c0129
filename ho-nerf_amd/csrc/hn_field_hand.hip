// Hand field: per-bone local coordinates (anerf_emb_point) -> 1386-wide masked encoding ->
// SDFNetwork + analytic d sdf / d p + RenderingNetwork, fused, one wave per 32 samples.
// The 1386 features are generated straight into MFMA B operands, bone by bone, and are
// never materialised.
//
// Reference: utils/fields.py:22-52 (bone coordinates), :132-177 (sdf net, .gradient),
// :222-240 (colour net), called from utils/renderer.py:137-142 / 390-396.
#include "hn_mlp.h"

namespace hn {

__constant__ float c_cutoff[N_BONES] = {0.08f, 0.03f, 0.03f, 0.02f, 0.02f, 0.03f, 0.02f, 0.02f, 0.02f, 0.03f, 0.02f,
                                        0.02f, 0.02f, 0.03f, 0.02f, 0.02f, 0.02f, 0.03f, 0.02f, 0.02f, 0.02f};
constexpr float TAU = 200.f;

struct FieldHandArgs {
    const float* pts;       // [n,3]
    const float* bt_inv;    // [n_frames,21,4,4]
    const float* T_pose;    // [n_frames,21,3]
    int n_pts;
    int pts_per_frame;
    int n_frames;
    // packed network
    const float4* w_fwd[9];   // l=0: over the bone-pair space [8][21*9]; 1..7 hidden; 8 feature rows
    const float4* w_skip;     // [8][21*9]
    const float* bias[9];
    const float* w8row;
    float b8;
    const float4* w_bwd[8];
    const float4* w_bwd_in0;  // [6 groups * 9 tiles][32]
    const float4* w_bwd_in4;
    const float4* c_in_x;     // [8][21*9]
    const float4* c_in_f;     // [8][32]
    const float4* c_in_g;     // [8][4]
    const float4* c_fwd[4];
    const float* c_bias[4];
    const float* c_wlast;
    float c_blast[3];
    float* sdf;
    float* grad;
    float* rgb;
    float* feat;
    float4* scratch;  // per-wave slots, see the HS_* enum
};

struct Bone {
    float v, r[3], hh;   // |q|, q/|q|, mask h
};

// utils/fields.py:26-35: q = R_b p + t_b - T_b; v = |q|; r = q/v; h = 1 - sigmoid(200 (v - cutoff_b))
__device__ __forceinline__ Bone bone_coords(const float p[3], const float* __restrict__ M, const float* __restrict__ Tp,
                                            int b) {
    Bone o;
    const float* m = M + 16 * b;
    const float q0 = m[0] * p[0] + m[1] * p[1] + m[2] * p[2] + m[3] - Tp[3 * b];
    const float q1 = m[4] * p[0] + m[5] * p[1] + m[6] * p[2] + m[7] - Tp[3 * b + 1];
    const float q2 = m[8] * p[0] + m[9] * p[1] + m[10] * p[2] + m[11] - Tp[3 * b + 2];
    o.v = sqrtf(q0 * q0 + q1 * q1 + q2 * q2);
    o.r[0] = q0 / o.v;     // no epsilon: a sample on a joint is NaN, as in the reference (SURVEY B-10)
    o.r[1] = q1 / o.v;
    o.r[2] = q2 / o.v;
    const float sg = 1.f / (1.f + expf(-TAU * (o.v - c_cutoff[b])));
    o.hh = 1.f - sg;
    return o;
}

// The 36 k-step B values of one bone for this lane (half 0: first member, half 1: second):
//  0: (v, r_x)  1: (r_y, r_z)  2..11: (sin, cos)(2^k v)  12..32: (sin, cos)(2^k r_c)  33..35: pad; all * h
__device__ __forceinline__ void bone_features(const Bone& bn, int h, float (&bf)[BONE_STEPS]) {
    bf[0] = (h ? bn.r[0] : bn.v) * bn.hh;
    bf[1] = (h ? bn.r[2] : bn.r[1]) * bn.hh;
    float f = 1.f;
#pragma unroll
    for (int k = 0; k < PTS_FREQS; ++k) {
        float s, c;
        sincos_acc(bn.v * f, &s, &c);
        bf[2 + k] = (h ? c : s) * bn.hh;
        f *= 2.f;
    }
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        f = 1.f;
#pragma unroll
        for (int k = 0; k < HAND_DIR_FREQS; ++k) {
            float s, c;
            sincos_acc(bn.r[ch] * f, &s, &c);
            bf[12 + HAND_DIR_FREQS * ch + k] = (h ? c : s) * bn.hh;
            f *= 2.f;
        }
    }
    bf[33] = 0.f;
    bf[34] = 0.f;
    bf[35] = 0.f;
}

__device__ __forceinline__ void encode_vec4_h(const float v[3], int h, float (&b)[VEC_STEPS]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float f = 1.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float s, co;
            sincos_acc(v[c] * f, &s, &co);
            b[c * 4 + k] = h ? co : s;
            f *= 2.f;
        }
    }
    b[12] = h ? v[1] : v[0];
    b[13] = h ? 0.f : v[2];
    b[14] = 0.f;
    b[15] = 0.f;
}

// Encoding Jacobian from the CACHED features.  A stored feature is phi * h; its partner lane
// (other half, same sample) stores the conjugate (cos for sin, sin for cos) * h, so
//   d(sin(f x) h)/dx = +f * partner + own * h'/h,   d(cos(f x) h)/dx = -f * partner + own * h'/h
// with h'/h = -tau * sigmoid(tau (v - cutoff)) =: kk -- no trigonometry is re-evaluated and
// nothing is divided by h.  Accumulates, for one bone,
//   Sv   = sum_i delta_i d(feature_i)/dv      Sr_c = sum_i delta_i d(feature_i)/dr_c
template <int PP>
__device__ __forceinline__ void jac_row(float delta, float own, const Bone& bn, float kk, int h, float& Sv,
                                        float (&Sr)[3]) {
    if constexpr (PP >= 33) {
        // pad rows carry zero weights
    } else if constexpr (PP == 0) {
        // (v h, r_x h)
        Sv += delta * own * kk;
        Sv += h ? 0.f : delta * bn.hh;
        Sr[0] += h ? delta * bn.hh : 0.f;
    } else if constexpr (PP == 1) {
        // (r_y h, r_z h)
        Sv += delta * own * kk;
        Sr[1] += h ? 0.f : delta * bn.hh;
        Sr[2] += h ? delta * bn.hh : 0.f;
    } else {
        const float other = __shfl_xor(own, 32, 64);
        if constexpr (PP < 12) {
            constexpr float f = (float)(1 << (PP - 2));
            Sv += delta * ((h ? -f : f) * other + own * kk);
        } else {
            constexpr int ch = (PP - 12) / HAND_DIR_FREQS;
            constexpr float f = (float)(1 << ((PP - 12) % HAND_DIR_FREQS));
            Sr[ch] += delta * (h ? -f : f) * other;
            Sv += delta * own * kk;
        }
    }
}

// One tile (16 pairs) of a 4-bone group: pairs 16 U .. 16 U + 15 of the group's 144.
// own[r] = this lane's cached feature of pair 16 U + r.
template <int U>
__device__ __forceinline__ void jac_tile(const f32x16& G, const float (&own)[16], const Bone (&bn)[BONE_GROUP],
                                         const float (&kk)[BONE_GROUP], int h, float (&Sv)[BONE_GROUP],
                                         float (&Sr)[BONE_GROUP][3]) {
#define HN_JROW(R)                                                                                         \
    {                                                                                                      \
        constexpr int P = 16 * U + (R);                                                                    \
        jac_row<P % BONE_STEPS>(G[R], own[R], bn[P / BONE_STEPS], kk[P / BONE_STEPS], h, Sv[P / BONE_STEPS], \
                                Sr[P / BONE_STEPS]);                                                       \
    }
    HN_JROW(0) HN_JROW(1) HN_JROW(2) HN_JROW(3) HN_JROW(4) HN_JROW(5) HN_JROW(6) HN_JROW(7)
    HN_JROW(8) HN_JROW(9) HN_JROW(10) HN_JROW(11) HN_JROW(12) HN_JROW(13) HN_JROW(14) HN_JROW(15)
#undef HN_JROW
}

// scratch slots of one wave (32 KiB each)
enum {
    HS_A1 = 0,        // a1..a8 -> slots 0..7
    HS_C0 = 8,        // colour lin0 pre-activation
    HS_DZ = 9,        // dz ping-pong 9, 10 (later: colour hidden activations)
    HS_DZ4 = 11,
    HS_FEAT = 12,     // cached bone features: [21 bones][9][64] float4 = 5.9 slots
    HAND_SLOTS_FULL = 18,
    HS_SDF_FEAT = 2,
    HAND_SLOTS_SDF = 8,
};

// acc[t] += W[t, bone block] * cached features, for all 21 bones (8 static accumulators)
__device__ __forceinline__ void feature_pass(f32x16 (&acc)[NT], const float4* __restrict__ w,
                                             const float4* __restrict__ feat_slot, int lane) {
#pragma unroll 1
    for (int b = 0; b < N_BONES; ++b) {
        float bf[BONE_STEPS];
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const float4 v = feat_slot[(b * 9 + q) * 64 + lane];
            bf[4 * q] = v.x;
            bf[4 * q + 1] = v.y;
            bf[4 * q + 2] = v.z;
            bf[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) mma_steps<9>(acc[t], w + ((size_t)t * N_BONES + b) * 9 * 64, bf, lane);
    }
}

// d sdf / d features = W^T dz for one of the two feature consumers (lin0 / lin4 skip), one
// 16-pair tile at a time, contracted with the encoding Jacobian; adds R_b^T d sdf/dq_b to g.
__device__ __forceinline__ void feature_grad_pass(const float4* __restrict__ wT, const float4* __restrict__ dz_slot,
                                                  const float4* __restrict__ feat_slot, const float p[3],
                                                  const float* __restrict__ M, const float* __restrict__ Tp, int lane,
                                                  int h, float (&g)[3]) {
    f32x16 x[NT];
    load_tiles<NT>(x, dz_slot, lane);
#pragma unroll 1
    for (int grp = 0; grp < N_GROUPS; ++grp) {
        Bone bn[BONE_GROUP];
        float kk[BONE_GROUP];
#pragma unroll
        for (int bi = 0; bi < BONE_GROUP; ++bi) {
            const int b = grp * BONE_GROUP + bi;
            bn[bi] = bone_coords(p, M, Tp, b < N_BONES ? b : N_BONES - 1);
            kk[bi] = -TAU * (1.f - bn[bi].hh);
        }
        float Sv[BONE_GROUP] = {0.f, 0.f, 0.f, 0.f};
        float Sr[BONE_GROUP][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
        const float4* w = wT + (size_t)grp * GROUP_TILES * 32 * 64;
        const float4* fs = feat_slot + (size_t)grp * BONE_GROUP * 9 * 64;
        const bool last = grp == N_GROUPS - 1;   // only bone 20 is real there: tiles 0..2
#define HN_GTILE(U)                                                                                     \
    if (!(last && (U) >= 3)) {                                                                          \
        f32x16 G = zero_tile();                                                                         \
        _Pragma("unroll") for (int u = 0; u < NT; ++u) mma_tile(G, w + ((size_t)(U) * NT + u) * 4 * 64, x[u], lane); \
        float own[16];                                                                                  \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                 \
            /* the last group has 1 real bone = 9 float4 groups; rows beyond it have zero weights */   \
            const int qi = 4 * (U) + q;                                                                 \
            const float4 v = (last && qi >= 9) ? make_float4(0.f, 0.f, 0.f, 0.f) : fs[qi * 64 + lane];  \
            own[4 * q] = v.x;                                                                           \
            own[4 * q + 1] = v.y;                                                                       \
            own[4 * q + 2] = v.z;                                                                       \
            own[4 * q + 3] = v.w;                                                                       \
        }                                                                                               \
        jac_tile<U>(G, own, bn, kk, h, Sv, Sr);                                                         \
    }
        HN_GTILE(0) HN_GTILE(1) HN_GTILE(2) HN_GTILE(3) HN_GTILE(4) HN_GTILE(5) HN_GTILE(6) HN_GTILE(7) HN_GTILE(8)
#undef HN_GTILE
#pragma unroll
        for (int bi = 0; bi < BONE_GROUP; ++bi) {
            const int b = grp * BONE_GROUP + bi;
            if (b >= N_BONES) continue;
            const float sv = half_sum(Sv[bi]);
            const float sr0 = half_sum(Sr[bi][0]), sr1 = half_sum(Sr[bi][1]), sr2 = half_sum(Sr[bi][2]);
            const Bone& q = bn[bi];
            // d/dq = Sv r + (Sr - (Sr.r) r) / v      (dv/dq = r, dr/dq = (I - r r^T)/v)
            const float dot = sr0 * q.r[0] + sr1 * q.r[1] + sr2 * q.r[2];
            const float dq0 = sv * q.r[0] + (sr0 - dot * q.r[0]) / q.v;
            const float dq1 = sv * q.r[1] + (sr1 - dot * q.r[1]) / q.v;
            const float dq2 = sv * q.r[2] + (sr2 - dot * q.r[2]) / q.v;
            const float* m = M + 16 * b;   // d/dp = R_b^T d/dq
            g[0] += m[0] * dq0 + m[4] * dq1 + m[8] * dq2;
            g[1] += m[1] * dq0 + m[5] * dq1 + m[9] * dq2;
            g[2] += m[2] * dq0 + m[6] * dq1 + m[10] * dq2;
        }
    }
}

template <bool FULL>
__global__ __launch_bounds__(64, 2) void k_field_hand(const FieldHandArgs a) {
    const int lane = threadIdx.x;
    const int j = lane & 31;
    const int h = lane >> 5;
    float4* const base = a.scratch + (size_t)blockIdx.x * (FULL ? HAND_SLOTS_FULL : HAND_SLOTS_SDF) * SLOT_FLOAT4;
    auto slot = [&](int i) { return base + (size_t)i * SLOT_FLOAT4; };
    auto act_slot = [&](int l) { return slot(FULL ? l - 1 : (l & 1)); };
    float4* const feat_slot = slot(FULL ? HS_FEAT : HS_SDF_FEAT);
    const int n_tiles = (a.n_pts + 31) / 32;

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int n = tile * 32 + j;
        const bool valid = n < a.n_pts;
        const int nn = valid ? n : a.n_pts - 1;
        const float p[3] = {a.pts[3 * nn], a.pts[3 * nn + 1], a.pts[3 * nn + 2]};
        int frame = nn / a.pts_per_frame;
        frame = frame < a.n_frames ? frame : a.n_frames - 1;
        const float* M = a.bt_inv + (size_t)frame * N_BONES * 16;
        const float* Tp = a.T_pose + (size_t)frame * N_BONES * 3;

        // ---- lin0 over the 21 x 36 bone pair-steps; the features are generated once, cached in
        //      the wave's scratch and re-read by lin4 (skip), colour lin0 and the Jacobian
        {
            f32x16 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = load_bias_tile(a.bias[0], t, h);
#pragma unroll 1
            for (int b = 0; b < N_BONES; ++b) {
                const Bone bn = bone_coords(p, M, Tp, b);
                float bf[BONE_STEPS];
                bone_features(bn, h, bf);
#pragma unroll
                for (int q = 0; q < 9; ++q)
                    feat_slot[(b * 9 + q) * 64 + lane] = make_float4(bf[4 * q], bf[4 * q + 1], bf[4 * q + 2], bf[4 * q + 3]);
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    mma_steps<9>(acc[t], a.w_fwd[0] + ((size_t)t * N_BONES + b) * 9 * 64, bf, lane);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                activate<ACT_SOFTPLUS>(acc[t]);
                store_tile(act_slot(1), t, acc[t], lane);
            }
        }
        layer_slots<NT, NT, ACT_SOFTPLUS>(a.w_fwd[1], a.bias[1], act_slot(1), act_slot(2), lane, h, NoExtra());
        layer_slots<NT, NT, ACT_SOFTPLUS>(a.w_fwd[2], a.bias[2], act_slot(2), act_slot(3), lane, h, NoExtra());
        layer_slots<NT, NT, ACT_SOFTPLUS>(a.w_fwd[3], a.bias[3], act_slot(3), act_slot(4), lane, h, NoExtra());
        // ---- lin4: hidden part (pre-activation parked in a5's slot), then the skip part over the features
        layer_slots<NT, NT, ACT_NONE>(a.w_fwd[4], a.bias[4], act_slot(4), act_slot(5), lane, h, NoExtra());
        {
            f32x16 acc[NT];
            load_tiles<NT>(acc, act_slot(5), lane);
            feature_pass(acc, a.w_skip, feat_slot, lane);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                activate<ACT_SOFTPLUS>(acc[t]);
                store_tile(act_slot(5), t, acc[t], lane);
            }
        }
        layer_slots<NT, NT, ACT_SOFTPLUS>(a.w_fwd[5], a.bias[5], act_slot(5), act_slot(6), lane, h, NoExtra());
        layer_slots<NT, NT, ACT_SOFTPLUS>(a.w_fwd[6], a.bias[6], act_slot(6), act_slot(7), lane, h, NoExtra());
        layer_slots<NT, NT, ACT_SOFTPLUS>(a.w_fwd[7], a.bias[7], act_slot(7), act_slot(8), lane, h, NoExtra());
        float sdf;
        {
            f32x16 x[NT];
            load_tiles<NT>(x, act_slot(8), lane);
            sdf = row_dot<NT>(a.w8row, x, h) + a.b8;
            if (!FULL) {
                if (valid && h == 0) a.sdf[n] = sdf;
                continue;
            }
            // lin8 rows 1..256 (feature vector) -> slot DZ
            layer_from_regs<NT, NT, ACT_NONE>(a.w_fwd[8], a.bias[8], x, slot(HS_DZ), lane, h, NoExtra());
        }
        if (a.feat != nullptr && valid) {
#pragma unroll 1
            for (int t = 0; t < NT; ++t) {
                const f32x16 f = load_tile(slot(HS_DZ), t, lane);
#pragma unroll
                for (int r = 0; r < 16; ++r) a.feat[(size_t)n * H + 32 * t + tile_row(r, h)] = f[r];
            }
        }
        // ---- colour lin0 without the gradient columns: [xyz_feature | feature vector] -> slot C0
        layer_slots<NT, NT, ACT_NONE>(a.c_in_f, a.c_bias[0], slot(HS_DZ), slot(HS_C0), lane, h, NoExtra());
        {
            f32x16 acc[NT];
            load_tiles<NT>(acc, slot(HS_C0), lane);
            feature_pass(acc, a.c_in_x, feat_slot, lane);
#pragma unroll
            for (int t = 0; t < NT; ++t) store_tile(slot(HS_C0), t, acc[t], lane);
        }
        // ---- reverse sweep: dz7 = sigma'(z7) * W8[0,:]
#pragma unroll 1
        for (int t = 0; t < NT; ++t) {
            const f32x16 act = load_tile(act_slot(8), t, lane);
            f32x16 dz;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 w = *reinterpret_cast<const float4*>(a.w8row + 32 * t + 8 * q + 4 * h);
                dz[4 * q + 0] = dsoftplus_from_act(act[4 * q + 0]) * w.x;
                dz[4 * q + 1] = dsoftplus_from_act(act[4 * q + 1]) * w.y;
                dz[4 * q + 2] = dsoftplus_from_act(act[4 * q + 2]) * w.z;
                dz[4 * q + 3] = dsoftplus_from_act(act[4 * q + 3]) * w.w;
            }
            store_tile(slot(HS_DZ + 1), t, dz, lane);   // dz7
        }
        layer_bwd_slots<NT, NT>(a.w_bwd[7], slot(HS_DZ + 1), act_slot(7), slot(HS_DZ + 0), lane);   // dz6
        layer_bwd_slots<NT, NT>(a.w_bwd[6], slot(HS_DZ + 0), act_slot(6), slot(HS_DZ + 1), lane);   // dz5
        layer_bwd_slots<NT, NT>(a.w_bwd[5], slot(HS_DZ + 1), act_slot(5), slot(HS_DZ4), lane);      // dz4
        layer_bwd_slots<NT, NT>(a.w_bwd[4], slot(HS_DZ4), act_slot(4), slot(HS_DZ + 1), lane);      // dz3
        layer_bwd_slots<NT, NT>(a.w_bwd[3], slot(HS_DZ + 1), act_slot(3), slot(HS_DZ + 0), lane);   // dz2
        layer_bwd_slots<NT, NT>(a.w_bwd[2], slot(HS_DZ + 0), act_slot(2), slot(HS_DZ + 1), lane);   // dz1
        layer_bwd_slots<NT, NT>(a.w_bwd[1], slot(HS_DZ + 1), act_slot(1), slot(HS_DZ + 0), lane);   // dz0
        // d sdf / d p through the features: W0^T dz0 and W4x^T dz4
        float g[3] = {0.f, 0.f, 0.f};
        feature_grad_pass(a.w_bwd_in0, slot(HS_DZ + 0), feat_slot, p, M, Tp, lane, h, g);
        feature_grad_pass(a.w_bwd_in4, slot(HS_DZ4), feat_slot, p, M, Tp, lane, h, g);
        // ---- colour: gradient columns, relu, lin1..lin3, lin4 + sigmoid
        {
            float bg[VEC_STEPS];
            encode_vec4_h(g, h, bg);
#pragma unroll 1
            for (int t = 0; t < NT; ++t) {
                f32x16 acc = load_tile(slot(HS_C0), t, lane);
                mma_steps<4>(acc, a.c_in_g + (size_t)t * 4 * 64, bg, lane);
                activate<ACT_RELU>(acc);
                store_tile(slot(HS_DZ), t, acc, lane);
            }
        }
        layer_slots<NT, NT, ACT_RELU>(a.c_fwd[1], a.c_bias[1], slot(HS_DZ), slot(HS_DZ + 1), lane, h, NoExtra());
        layer_slots<NT, NT, ACT_RELU>(a.c_fwd[2], a.c_bias[2], slot(HS_DZ + 1), slot(HS_DZ), lane, h, NoExtra());
        layer_slots<NT, NT, ACT_RELU>(a.c_fwd[3], a.c_bias[3], slot(HS_DZ), slot(HS_DZ + 1), lane, h, NoExtra());
        float rgb[3];
        {
            f32x16 x[NT];
            load_tiles<NT>(x, slot(HS_DZ + 1), lane);
#pragma unroll
            for (int c = 0; c < 3; ++c) rgb[c] = sigmoidf_(row_dot<NT>(a.c_wlast + c * H, x, h) + a.c_blast[c]);
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

constexpr int FIELD_WAVES_PER_CU = 8;
static int field_grid(int n_pts, int n_cus) {
    const int n_tiles = (n_pts + 31) / 32;
    const int cap = n_cus * FIELD_WAVES_PER_CU;
    return n_tiles < cap ? n_tiles : cap;
}

size_t field_hand_workspace_bytes(int n_pts, int n_cus) {
    return (size_t)field_grid(n_pts, n_cus) * HAND_SLOTS_FULL * SLOT_FLOAT4 * sizeof(float4);
}

int launch_field_hand(const hn_field* f, const float* pts, int n_pts, const float* bt_inv, const float* T_pose,
                      int n_frames, int pts_per_frame, float* sdf, float* grad, float* rgb, float* feat,
                      void* workspace, size_t workspace_bytes, bool full, hipStream_t stream) {
    if (n_pts <= 0) return HN_OK;
    HN_REQUIRE(bt_inv != nullptr && T_pose != nullptr && n_frames >= 1 && pts_per_frame >= 1,
               "hand field needs bt_inv / T_pose and frame sizes");
    FieldHandArgs a;
    a.pts = pts;
    a.bt_inv = bt_inv;
    a.T_pose = T_pose;
    a.n_pts = n_pts;
    a.pts_per_frame = pts_per_frame;
    a.n_frames = n_frames;
    for (int l = 0; l < 9; ++l) {
        a.w_fwd[l] = f->sdf_fwd[l].w;
        a.bias[l] = f->sdf_bias[l];
    }
    a.w_skip = f->sdf_skip.w;
    a.w8row = f->sdf_w8row;
    a.b8 = f->sdf_b8;
    for (int l = 0; l < 8; ++l) a.w_bwd[l] = f->sdf_bwd[l].w;
    a.w_bwd_in0 = f->sdf_bwd_in0.w;
    a.w_bwd_in4 = f->sdf_bwd_in4.w;
    a.c_in_x = f->col_in_x.w;
    a.c_in_f = f->col_in_f.w;
    a.c_in_g = f->col_in_g.w;
    for (int l = 0; l < 4; ++l) {
        a.c_fwd[l] = f->col_fwd[l].w;
        a.c_bias[l] = f->col_bias[l];
    }
    a.c_wlast = f->col_wlast;
    for (int c = 0; c < 3; ++c) a.c_blast[c] = f->col_blast[c];
    a.sdf = sdf;
    a.grad = grad;
    a.rgb = rgb;
    a.feat = feat;
    a.scratch = reinterpret_cast<float4*>(workspace);
    int n_cus = device_cus();
    if (n_cus <= 0) n_cus = 256;
    const int grid = field_grid(n_pts, n_cus);
    const size_t need = (size_t)grid * (full ? HAND_SLOTS_FULL : HAND_SLOTS_SDF) * SLOT_FLOAT4 * sizeof(float4);
    if (workspace == nullptr || workspace_bytes < need) {
        set_error("field workspace too small: %zu < %zu", workspace_bytes, need);
        return HN_ENOMEM;
    }
    if (full)
        hipLaunchKernelGGL(k_field_hand<true>, dim3(grid), dim3(64), 0, stream, a);
    else
        hipLaunchKernelGGL(k_field_hand<false>, dim3(grid), dim3(64), 0, stream, a);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

}  // namespace hn
