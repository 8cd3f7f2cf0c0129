// v2 object field (f16x3 MFMA, LDS-streamed weights): SDFNetwork_OBJ forward, analytic
// d sdf / d p (reverse sweep) and RenderingNetwork_OBJ, fused; 4 waves x 32 samples per
// workgroup, one workgroup per CU, activations in registers (hn_mlp2.h).
//
// Reference: utils/fields.py:316-347 (sdf net, .gradient), :387-405 (colour net), called from
// utils/renderer.py:130-135 / 380-385.  The chunk order below is the contract with
// hn_pack2.hip (build_obj_stream).
#include <stdlib.h>

#include <atomic>

#include "hn_mlp2.h"
#ifndef HN_PARK_AGPR
#define HN_PARK_AGPR 1
#endif

namespace hn {
namespace v2 {

struct Obj2Args {
    const float* pts;      // [n,3]
    const float* rays_d;   // [n/spr,3]
    int n_pts;
    int spr;
    float inv_scale;
    const char* blob;      // weight stream (FULL or SDF-only program)
    size_t blob_bytes;
    const float* b8;   // &bias of lin8's sdf row (the field's retained copy, read on the device)
    const float* c_blast;   // the three biases of colour lin4 (the field's retained copy, read on the device)
    float* sdf;
    float* grad;
    float* rgb;
    float* feat;           // optional [n,256]
    float4* scratch;       // per-wave stash slots (FULL only)
    int dbg;               // timing experiments only (HN_DBG): 1 = no stash stores, 2 = no stash loads
    unsigned* xsync;       // XCD pacing counters (hn_mlp2.h XcdPace) or NULL
    // adjoint (MODE 2): upstream gradients in, input gradients out
    const float* g_sdf;    // [n]
    const float* g_grad;   // [n,3]
    const float* g_rgb;    // [n,3]
    float* g_pts;          // [n,3]
    float* g_rays_d;       // [n/spr,3] or NULL: accumulated with atomics (zeroed by the launcher); with dir_per_sample: [n,3], stored
    int dir_per_sample;
    // MODE 5 (the adjoint from a tape that also leaves the PER-LAYER SIGNALS of the parameter gradients): OSG_COUNT row-major
    // [n, 256] fp32 arrays `sig + k * sig_pitch` (enum below), unscaled, and gb [n,3] (the adjoint of d sdf / d pts incl. the colour
    // network's share: J gb is the forward-direction sweep's input)
    float* sig;
    size_t sig_pitch;      // floats between two signal arrays
    float* gb_out;
};
// signal arrays of MODE 5 (what hn_field_bwd.hip's outer products pair up: zb_l (x) a_l, dz_l (x) v_{l-1}, cb_l (x) c_l, fb (x) a_8):
//   OSG_CB + k: adjoint of colour layer (3 - k)'s pre-activation;  OSG_C + k: c_{k+1} (colour activations);  OSG_A + l: a_{l+1};
//   OSG_DZ + l: dz_l of the reverse sweep;  OSG_V + l: v_l = sigma'_l dzb_l (forward-direction sweep);  OSG_ZB + l: zb_l (second reverse
//   sweep);  OSG_FB: the feature vector's adjoint
enum { OSG_CB = 0, OSG_C = 4, OSG_A = 8, OSG_DZ = 16, OSG_V = 24, OSG_ZB = 32, OSG_FB = 40, OSG_COUNT = 41 };

// stash slots of one wave (32 KiB each)
enum { OS_A1 = 0 /* a1..a7 -> 0..6 */, OS_DZ7 = 7, OS_FVEC = 8, OS_DZ4 = 9, OS_X = 10, OBJ2_SLOTS = 11 };
// ... and what the adjoint adds: a8, the reverse sweep's dz_l as fp32 tiles (slot l; each is overwritten by the
// second-order source w_l once the forward-direction sweep has passed layer l), the colour network's activations
// (for the relu masks), d sdf / d X (tiles 0, 1) with the colour net's share of the X adjoint (tiles 2, 3), zb4
enum { OS_A8 = 11, OS_DZ = 12 /* 12..19 */, OS_C = 20 /* c1..c4 -> 20..23 */, OS_GX = 24, OS_ZB4 = 25, OBJ2_SLOTS_ADJ = 26 };

constexpr int CB_HID = chunk_bytes(1, 16, true);     // hidden layer tile: 16 k-steps + tail
constexpr int CB_L0 = chunk_bytes(4, 4, true);       // lin0: 4 tiles x 4 k-steps + tail
constexpr int CB_BWD = chunk_bytes(1, 16, false);    // transposed hidden tile
constexpr int CB_BWD3 = chunk_bytes(1, 13, false);   // W3^T: 193 outputs = 13 k-steps
constexpr int CB_C0A = chunk_bytes(1, 16, false);    // colour lin0, feature-vector columns
constexpr int CB_C0B = chunk_bytes(1, 8, true);      // colour lin0, enc(p) | enc(d) | enc(g) columns + bias
constexpr int CB_W4ROWS = 3 * TAIL_BYTES;            // adjoint: the three rows of colour lin4, one tail-format KiB each

// sin/cos(2^k x) of one lane half: half 0 keeps the sines, half 1 the cosines
__device__ __forceinline__ float sc_half(float ang, int h) {
    float s, c;
    sincos_cw(ang, s, c);
    return h ? c : s;
}
// X space of the sdf net, 4 k-steps x 8 values per lane (see build_obj_stream: x_slots)
__device__ __forceinline__ void encode_x(const float p[3], int h, float (&f)[4][8]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float fr = 1.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            f[c][k] = sc_half(p[c] * fr, h);
            fr *= 2.f;
        }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) f[3][j] = sc_half(p[j >> 1] * ((j & 1) ? 512.f : 256.f), h);
    f[3][6] = h ? p[2] : p[0];
    f[3][7] = h ? 0.f : p[1];
}
// enc4 of a 3-vector, 2 k-steps x 8 values per lane (see build_obj_stream: vec_slots)
__device__ __forceinline__ void encode_v4(const float v[3], int h, float (&f)[2][8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) f[0][j] = sc_half(v[j >> 2] * (float)(1 << (j & 3)), h);
#pragma unroll
    for (int j = 0; j < 4; ++j) f[1][j] = sc_half(v[2] * (float)(1 << j), h);
    f[1][4] = h ? v[2] : v[0];
    f[1][5] = h ? 0.f : v[1];
    f[1][6] = 0.f;
    f[1][7] = 0.f;
}


// ---- adjoint of the evaluation above (MODE 2) --------------------------------------------------------------------
// oracle/field_bwd.py steps 3b-6 for the tile whose forward pass has just finished (its tape is in the wave's stash):
//   colour network backward -> (Xb_c, d-bar, fb, gb)         [C3^T, C2^T, C1^T, C0^T chunks]
//   forward-direction sweep  dzb_l = W_l (sigma'_{l-1} dzb_{l-1}),  w_l = sigma''(z_l) u_l dzb_l   [the forward chunks again]
//   second reverse sweep     zb_l = sigma'(z_l) ab_l + w_l,  ab_{l-1} = W_l^T zb_l,  Xb += W0^T zb0 + W4x^T zb4
//   input map                g_pts = J^T Xb + (d^2 X / dp^2 : GX) gb
// Every adjoint quantity is linear in the upstream gradients (g_sdf, g_grad, g_rgb); they are scaled per sample by a
// power of two kappa that brings the largest of them to [1, 2), so that the fp16 hi/lo fragments keep their 22 bits
// whatever the scale of the caller's loss; the outputs are scaled back by 1 / kappa (exact).
struct Act2 {
    f32x16 v;   // stashed activation a_{l+1} (sigma'(z_l) = 1 - exp(-100 a))
    f32x16 x;   // second tile: dz_l (forward-direction sweep; kind 4 applies 100 / 256) or w_l (second reverse sweep)
};
struct Act1 {
    f32x16 v;
};
template <bool PG>
__device__ __forceinline__ void obj_adjoint(const Obj2Args& a, WStream& ws, Stash& sh, int lane, int h, int n, int nn, bool valid,
                                            int next_first /* first chunk of the next tile's program, 0 = none */, const float (&p)[3], const float (&d)[3], const float (&g)[3],
                                            const float (&rgb)[3], h8 (&ah)[16], h8 (&al)[16], h8 (&bh)[16], h8 (&bl)[16]) {
    auto no_store = [](auto, const auto&) {};
    auto no_pre = [](auto, const char*) { return NoData{}; };
    // MODE 5: tile t of a [neurons x samples] accumulator tile -> columns 32 t .. 32 t + 31 of this lane's sample row of signal array
    // `arr` (register i of lane (h, j): neuron 8 (i / 4) + 4 h + (i % 4), sample j), times `scale` (powers of two: exact)
    float sig_scale = 1.f;   // 1 / kappa once it is known (below)
    auto sig = [&](int arr, int t, const f32x16& y, float scale) {
        if constexpr (PG) {
            if (valid) {
                using f32x4 = float __attribute__((ext_vector_type(4)));
                float* row = a.sig + (size_t)arr * a.sig_pitch + (size_t)n * 256 + 32 * t + 4 * h;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
                    *reinterpret_cast<f32x4*>(row + 8 * g4) = f32x4{y[4 * g4] * scale, y[4 * g4 + 1] * scale, y[4 * g4 + 2] * scale, y[4 * g4 + 3] * scale};
            }
        }
    };
    auto sig1 = [&](int arr, int col, float v) {   // one column (lin3's neuron 192: register 0 of tile 6, half 0)
        if constexpr (PG) {
            if (valid && h == 0) a.sig[(size_t)arr * a.sig_pitch + (size_t)n * 256 + col] = v;
        }
    };
    auto to_regs = [&](h8(&oh)[16], h8(&ol)[16]) {
        return [&oh, &ol](auto T, EpiState& st, const auto&) {
            constexpr int t = decltype(T)::value;
            asm volatile("" : "+v"(st.hi[0]), "+v"(st.lo[0]), "+v"(st.hi[1]), "+v"(st.lo[1]));
            oh[2 * t] = st.hi[0];
            ol[2 * t] = st.lo[0];
            oh[2 * t + 1] = st.hi[1];
            ol[2 * t + 1] = st.lo[1];
#if HN_PARK_AGPR
            asm volatile("" : "+a"(oh[2 * t]), "+a"(ol[2 * t]), "+a"(oh[2 * t + 1]), "+a"(ol[2 * t + 1]));
#endif
            return NoData{};
        };
    };
    // ... the same, and the tile's value -> signal array `arr` (times 1 / kappa), the pre-data's first tile -> `arr_pre` (as it is)
    auto to_regs_sig = [&](h8(&oh)[16], h8(&ol)[16], int arr, int arr_pre) {
        return [&oh, &ol, arr, arr_pre, &sig, &sig_scale](auto T, EpiState& st, const auto& pd) {
            constexpr int t = decltype(T)::value;
            asm volatile("" : "+v"(st.hi[0]), "+v"(st.lo[0]), "+v"(st.hi[1]), "+v"(st.lo[1]));
            oh[2 * t] = st.hi[0];
            ol[2 * t] = st.lo[0];
            oh[2 * t + 1] = st.hi[1];
            ol[2 * t + 1] = st.lo[1];
#if HN_PARK_AGPR
            asm volatile("" : "+a"(oh[2 * t]), "+a"(ol[2 * t]), "+a"(oh[2 * t + 1]), "+a"(ol[2 * t + 1]));
#endif
            if constexpr (PG) {
                sig(arr, t, st.vec(), sig_scale);
                if (arr_pre >= 0) sig(arr_pre, t, pd.v, 1.f);
            }
            return NoData{};
        };
    };
    auto mask_of = [&](int c_slot) {
        return [&sh, c_slot](auto T, const char*) { return Act1{sh.tile_load(c_slot, decltype(T)::value)}; };
    };
    // ---- seeds
    float gs = a.g_sdf[nn], gg[3], gr[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        gg[c] = a.g_grad[3 * nn + c];
        gr[c] = a.g_rgb[3 * nn + c];
    }
    if (!valid) {   // lanes beyond the end shadow the last sample: they must not contribute
        gs = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) gg[c] = gr[c] = 0.f;
    }
    float kappa = 1.f, inv_kappa = 1.f;
    {
        float m = fabsf(gs);
#pragma unroll
        for (int c = 0; c < 3; ++c) m = fmaxf(m, fmaxf(fabsf(gg[c]), fabsf(gr[c])));
        const unsigned e = (__builtin_bit_cast(unsigned, m) >> 23) & 0xffu;   // m in [2^(e-127), 2^(e-126))
        if (e >= 1u && e <= 253u) {
            kappa = __builtin_bit_cast(float, (254u - e) << 23);
            inv_kappa = __builtin_bit_cast(float, e << 23);
        }
    }
    sig_scale = inv_kappa;
    const float gsk = gs * kappa * a.inv_scale;
    float xb[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) xb[c] = kappa * gr[c] * rgb[c] * (1.f - rgb[c]);

    // ---- colour lin4^T and the mask of c4:  cb4 = (c4 > 0) * (W_c4^T xb)   (the three rows arrive as one small chunk)
    {
        const char* buf = ws.template acquire<0>();
        ws.template begin_c<CB_BWD>();
        ws.template pieces_all_c<CB_BWD>();
        static_for<8>([&](auto T) {
            constexpr int t = decltype(T)::value;
            const f32x16 c4 = sh.tile_load(OS_C + 3, t);
            const f32x16 w0 = tail_tile(buf, t, h), w1 = tail_tile(buf + TAIL_BYTES, t, h), w2 = tail_tile(buf + 2 * TAIL_BYTES, t, h);
            f32x16 v;
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = c4[i] > 0.f ? fmaf(w0[i], xb[0], fmaf(w1[i], xb[1], w2[i] * xb[2])) : 0.f;
            split_tile(v, ah[2 * t], al[2 * t], ah[2 * t + 1], al[2 * t + 1]);
            sig(OSG_CB + 0, t, v, sig_scale);
            sig(OSG_C + 3, t, c4, 1.f);
        });
    }
    run_layer_c<8, 16, 1, false, true, CB_BWD, CB_BWD>(ws, ah, al, lane, h, mask_of(OS_C + 2), PhMask{}, to_regs_sig(bh, bl, OSG_CB + 1, OSG_C + 2), no_store);   // C3^T -> cb3
    run_layer_c<8, 16, 1, false, true, CB_BWD, CB_BWD>(ws, bh, bl, lane, h, mask_of(OS_C + 1), PhMask{}, to_regs_sig(ah, al, OSG_CB + 2, OSG_C + 1), no_store);   // C2^T -> cb2
    run_layer_c<8, 16, 1, false, true, CB_BWD, CB_BWD>(ws, ah, al, lane, h, mask_of(OS_C + 0), PhMask{}, to_regs_sig(bh, bl, OSG_CB + 3, OSG_C + 0), no_store);   // C1^T -> cb1
    // ---- colour lin0^T: feature-vector rows -> fb (kept as fragments in the OS_FVEC slot for the W8 product) ...
    run_layer_c<8, 16, 1, false, true, CB_BWD, CB_BWD>(
        ws, bh, bl, lane, h, no_pre, PhIdentity{},
        [&](auto T, EpiState& st, const auto&) {
            constexpr int t = decltype(T)::value;
            sh.frag_store(OS_FVEC * SLOT_BYTES, 2 * t, st.hi[0], st.lo[0]);
            sh.frag_store(OS_FVEC * SLOT_BYTES, 2 * t + 1, st.hi[1], st.lo[1]);
            sig(OSG_FB, t, st.vec(), sig_scale);
            return NoData{};
        },
        no_store);
    // ... and the [X | enc(d) | enc(g)] slots: 4 tiles whose row r of tile u is k-slot (s = 2u + (reg >> 3), h, reg & 7)
    // of the same lane.  Tiles 0, 1: the colour net's share of the X adjoint (parked); 2: enc(d); 3: enc(g).
    float gdir[3] = {0.f, 0.f, 0.f}, gb[3] = {0.f, 0.f, 0.f};
    static_for<4>([&](auto U) {
        constexpr int u = decltype(U)::value;
        const char* buf = ws.template acquire<0>();
        constexpr int nbytes = u < 3 ? CB_BWD : CB_L0;   // after the last one: lin0 of the forward-direction sweep
        ws.template begin_c<nbytes>();
        f32x16 m1 = zero16(), m2 = zero16();
        mma_tile<16, 0, nbytes>(ws, buf, bh, bl, m1, m2, lane);
        const f32x16 M = combine(m1, m2);
        if constexpr (u < 2) {
            sh.tile_store(OS_GX, 2 + u, M);
        } else {
            // J^T of [v, enc4(v)] in-lane: slot (s, j) of k-steps (0, 1) is register 8 s + j; the partner lane (other half)
            // holds the conjugate function of the same angle
            const float* vec = u == 2 ? d : g;
            float* out = u == 2 ? gdir : gb;
            float f[2][8];
            encode_v4(vec, h, f);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float fr = (float)(1 << (j & 3));
                out[j >> 2] = fmaf(M[j], (h ? -fr : fr) * other_half(f[0][j], h), out[j >> 2]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float fr = (float)(1 << j);
                out[2] = fmaf(M[8 + j], (h ? -fr : fr) * other_half(f[1][j], h), out[2]);
            }
            out[0] += h ? 0.f : M[12];
            out[2] += h ? M[12] : 0.f;
            out[1] += h ? 0.f : M[13];
#pragma unroll
            for (int c = 0; c < 3; ++c) out[c] = half_sum(out[c]);
        }
    });
#pragma unroll
    for (int c = 0; c < 3; ++c) gb[c] += kappa * gg[c];   // gb = g_grad + J_enc^T (colour net's gradient w.r.t. enc(g))
    if constexpr (PG) {
        if (valid && h == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) a.gb_out[3 * (size_t)n + c] = gb[c] * inv_kappa;
        }
    }

    // ---- forward-direction sweep: GXb = J gb -> dzb_0 = W0 GXb -> ... ; v_l = sigma'_l dzb_l feeds the next layer,
    //      w_l = (1 - sigma'_l) dzb_l * 100 dz_l replaces dz_l in the stash
    auto pre4 = [&](int act_slot, int dz_slot) {
        return [&sh, act_slot, dz_slot](auto T, const char*) {
            constexpr int t = decltype(T)::value;
            return Act2{sh.tile_load(act_slot, t), sh.tile_load(dz_slot, t)};   // (x = dz_l; kind 4 applies its 100 / 256)
        };
    };
    auto fin4 = [&](h8(&oh)[16], h8(&ol)[16], int w_slot) {
        return [&oh, &ol, w_slot, &sh, &sig, &sig_scale](auto T, EpiState& st, const auto& pd) {
            constexpr int t = decltype(T)::value;
            if constexpr (PG) {   // layer l = w_slot - OS_DZ: v_l, a_{l+1}, dz_l
                sig(OSG_V + (w_slot - OS_DZ), t, st.vec(), sig_scale);
                sig(OSG_A + (w_slot - OS_DZ), t, pd.v, 1.f);
                sig(OSG_DZ + (w_slot - OS_DZ), t, pd.x, BWD_INV);
            }
            asm volatile("" : "+v"(st.hi[0]), "+v"(st.lo[0]), "+v"(st.hi[1]), "+v"(st.lo[1]));
            oh[2 * t] = st.hi[0];
            ol[2 * t] = st.lo[0];
            oh[2 * t + 1] = st.hi[1];
            ol[2 * t + 1] = st.lo[1];
#if HN_PARK_AGPR
            asm volatile("" : "+a"(oh[2 * t]), "+a"(ol[2 * t]), "+a"(oh[2 * t + 1]), "+a"(ol[2 * t + 1]));
#endif
            sh.tile_store(w_slot, t, st.wvec());
            return NoData{};
        };
    };
    {
        // X-space fragments of J gb: slot value = d(slot function)/dp_c * gb[c]
        float f[4][8], jg[4][8];
        encode_x(p, h, f);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float fr = 1.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                jg[c][k] = (h ? -fr : fr) * other_half(f[c][k], h) * gb[c];
                fr *= 2.f;
            }
        }
#pragma unroll
        for (int jj = 0; jj < 6; ++jj) {
            const float fr = (jj & 1) ? 512.f : 256.f;
            jg[3][jj] = (h ? -fr : fr) * other_half(f[3][jj], h) * gb[jj >> 1];
        }
        jg[3][6] = h ? gb[2] : gb[0];
        jg[3][7] = h ? 0.f : gb[1];
        h8 x16h[16], x16l[16];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            split8(jg[s], x16h[s], x16l[s]);
            sh.frag_store(OS_X * SLOT_BYTES, s, x16h[s], x16l[s]);   // again for lin4's skip columns
        }
        run_layer_c<8, 4, 4, false, true, CB_L0, CB_HID>(ws, x16h, x16l, lane, h, pre4(OS_A1 + 0, OS_DZ + 0), PhFwdDir{}, fin4(ah, al, OS_DZ + 0), no_store);
    }
    run_layer_c<8, 16, 1, false, true, CB_HID, CB_HID>(ws, ah, al, lane, h, pre4(OS_A1 + 1, OS_DZ + 1), PhFwdDir{}, fin4(bh, bl, OS_DZ + 1), no_store);   // lin1
    run_layer_c<8, 16, 1, false, true, CB_HID, CB_HID>(ws, bh, bl, lane, h, pre4(OS_A1 + 2, OS_DZ + 2), PhFwdDir{}, fin4(ah, al, OS_DZ + 2), no_store);   // lin2
    float v3_192 = 0.f;
    run_layer_c<7, 16, 1, false, true, CB_HID, CB_HID>(ws, ah, al, lane, h, pre4(OS_A1 + 3, OS_DZ + 3), PhFwdDir{},                                    // lin3 (193 rows)
                                     [&](auto T, EpiState& st, const auto& pd) {
                                         constexpr int t = decltype(T)::value;
                                         asm volatile("" : "+v"(st.hi[0]), "+v"(st.lo[0]), "+v"(st.hi[1]), "+v"(st.lo[1]));
                                         if constexpr (PG) {
                                             if constexpr (t == 6) {
                                                 sig1(OSG_V + 3, 192, st.v[0] * sig_scale);
                                                 sig1(OSG_A + 3, 192, pd.v[0]);
                                                 sig1(OSG_DZ + 3, 192, pd.x[0] * BWD_INV);
                                             } else {
                                                 sig(OSG_V + 3, t, st.vec(), sig_scale);
                                                 sig(OSG_A + 3, t, pd.v, 1.f);
                                                 sig(OSG_DZ + 3, t, pd.x, BWD_INV);
                                             }
                                         }
                                         if constexpr (t == 6) {
                                             v3_192 = st.v[0];   // row 0 of tile 6 = neuron 192 (half 0); the padding rows are 0 (sigma' = 0)
                                         } else {
                                             bh[2 * t] = st.hi[0];
                                             bl[2 * t] = st.lo[0];
                                             bh[2 * t + 1] = st.hi[1];
                                             bl[2 * t + 1] = st.lo[1];
                                         }
                                         sh.tile_store(OS_DZ + 3, t, st.wvec());
                                         return NoData{};
                                     },
                                     no_store);
    {   // lin4 = [v3 (192 via k-steps 0..11) | J gb with v3[192] in its pad slot] / sqrt2
        const float v192 = other_half(h ? 0.f : v3_192, h);
#pragma unroll
        for (int s = 0; s < 4; ++s) sh.frag_load(OS_X * SLOT_BYTES, s, bh[12 + s], bl[12 + s]);
        const _Float16 vh = hi_part(v192);
        const _Float16 vl = (_Float16)((v192 - (float)vh) * LO_SCALE);
        bh[15][7] = h ? vh : bh[15][7];
        bl[15][7] = h ? vl : bl[15][7];
    }
    run_layer_c<8, 16, 1, false, true, CB_HID, CB_HID>(ws, bh, bl, lane, h, pre4(OS_A1 + 4, OS_DZ + 4), PhFwdDir{}, fin4(ah, al, OS_DZ + 4), no_store);   // lin4
    run_layer_c<8, 16, 1, false, true, CB_HID, CB_HID>(ws, ah, al, lane, h, pre4(OS_A1 + 5, OS_DZ + 5), PhFwdDir{}, fin4(bh, bl, OS_DZ + 5), no_store);   // lin5
    run_layer_c<8, 16, 1, false, true, CB_HID, CB_HID>(ws, bh, bl, lane, h, pre4(OS_A1 + 6, OS_DZ + 6), PhFwdDir{}, fin4(ah, al, OS_DZ + 6), no_store);   // lin6
    run_layer_c<8, 16, 1, false, false, CB_HID, CB_HID>(ws, ah, al, lane, h, pre4(OS_A8, OS_DZ + 7), PhFwdDir{},                                       // lin7: only w_7
                                      [&](auto T, EpiState& st, const auto& pd) {
                                          if constexpr (PG) {
                                              sig(OSG_V + 7, decltype(T)::value, st.vec(), sig_scale);
                                              sig(OSG_A + 7, decltype(T)::value, pd.v, 1.f);
                                              sig(OSG_DZ + 7, decltype(T)::value, pd.x, BWD_INV);
                                          }
                                          sh.tile_store(OS_DZ + 7, decltype(T)::value, st.wvec());
                                          return NoData{};
                                      },
                                      no_store);

    // ---- second reverse sweep.  ab_7 = W8[1:, :]^T fb + g_sdf / scale * W8[0, :];  zb_7 = sigma'_7 ab_7 + w_7
    auto pre5 = [&](int act_slot, int w_slot) {
        return [&sh, act_slot, w_slot](auto T, const char*) {
            constexpr int t = decltype(T)::value;
            return Act2{sh.tile_load(act_slot, t), sh.tile_load(w_slot, t)};
        };
    };
#pragma unroll
    for (int s = 0; s < 16; ++s) sh.frag_load(OS_FVEC * SLOT_BYTES, s, bh[s], bl[s]);
    run_layer_c<8, 16, 1, false, true, CB_HID, CB_BWD>(
        ws, bh, bl, lane, h,
        [&](auto T, const char* tail) {
            constexpr int t = decltype(T)::value;
            Act2 o{sh.tile_load(OS_A8, t), sh.tile_load(OS_DZ + 7, t)};
            const f32x16 w8 = tail_tile(tail, 0, h);
#pragma unroll
            for (int i = 0; i < 16; ++i) o.x[i] = fmaf(gsk * w8[i], dsoftplus_from_act(o.v[i]), o.x[i]);   // + sigma'_7 g_sdf W8[0, :]
            return o;
        },
        PhRev2{}, to_regs_sig(ah, al, OSG_ZB + 7, -1), no_store);
    run_layer_c<8, 16, 1, false, true, CB_BWD, CB_BWD>(ws, ah, al, lane, h, pre5(OS_A1 + 6, OS_DZ + 6), PhRev2{}, to_regs_sig(bh, bl, OSG_ZB + 6, -1), no_store);   // W7^T -> zb6
    run_layer_c<8, 16, 1, false, true, CB_BWD, CB_BWD>(ws, bh, bl, lane, h, pre5(OS_A1 + 5, OS_DZ + 5), PhRev2{}, to_regs_sig(ah, al, OSG_ZB + 5, -1), no_store);   // W6^T -> zb5
    run_layer_c<8, 16, 1, false, true, CB_BWD, CB_BWD>(ws, ah, al, lane, h, pre5(OS_A1 + 4, OS_DZ + 4), PhRev2{},                               // W5^T -> zb4 (kept)
                                     [&](auto T, EpiState& st, const auto&) {
                                         constexpr int t = decltype(T)::value;
                                         asm volatile("" : "+v"(st.hi[0]), "+v"(st.lo[0]), "+v"(st.hi[1]), "+v"(st.lo[1]));
                                         bh[2 * t] = st.hi[0];
                                         bl[2 * t] = st.lo[0];
                                         bh[2 * t + 1] = st.hi[1];
                                         bl[2 * t + 1] = st.lo[1];
                                         sh.frag_store(OS_ZB4 * SLOT_BYTES, 2 * t, st.hi[0], st.lo[0]);
                                         sh.frag_store(OS_ZB4 * SLOT_BYTES, 2 * t + 1, st.hi[1], st.lo[1]);
                                         sig(OSG_ZB + 4, t, st.vec(), sig_scale);
                                         return NoData{};
                                     },
                                     no_store);
    run_layer_c<7, 16, 1, false, true, CB_BWD, CB_BWD3>(ws, bh, bl, lane, h, pre5(OS_A1 + 3, OS_DZ + 3), PhRev2{},                              // W4h^T -> zb3 (193 rows)
                                     [&](auto T, EpiState& st, const auto&) {
                                         constexpr int t = decltype(T)::value;
                                         asm volatile("" : "+v"(st.hi[0]), "+v"(st.lo[0]), "+v"(st.hi[1]), "+v"(st.lo[1]));
                                         if constexpr (t < 6) {
                                             ah[2 * t] = st.hi[0];
                                             al[2 * t] = st.lo[0];
                                             ah[2 * t + 1] = st.hi[1];
                                             al[2 * t + 1] = st.lo[1];
                                             sig(OSG_ZB + 3, t, st.vec(), sig_scale);
                                         } else {
                                             ah[12] = st.hi[0];
                                             al[12] = st.lo[0];
                                             sig1(OSG_ZB + 3, 192, st.v[0] * sig_scale);
                                         }
                                         return NoData{};
                                     },
                                     no_store);
    run_layer_c<8, 13, 1, false, true, CB_BWD3, CB_BWD>(ws, ah, al, lane, h, pre5(OS_A1 + 2, OS_DZ + 2), PhRev2{}, to_regs_sig(bh, bl, OSG_ZB + 2, -1), no_store);   // W3^T -> zb2
    run_layer_c<8, 16, 1, false, true, CB_BWD, CB_BWD>(ws, bh, bl, lane, h, pre5(OS_A1 + 1, OS_DZ + 1), PhRev2{}, to_regs_sig(ah, al, OSG_ZB + 1, -1), no_store);    // W2^T -> zb1
    run_layer_c<8, 16, 1, false, true, CB_BWD, CB_BWD>(ws, ah, al, lane, h, pre5(OS_A1 + 0, OS_DZ + 0), PhRev2{}, to_regs_sig(bh, bl, OSG_ZB + 0, -1), no_store);    // W1^T -> zb0
    // X adjoint = W0^T zb0 + W4[:, 193:]^T zb4 + the colour net's share
    f32x16 G1[2] = {zero16(), zero16()}, G2[2] = {zero16(), zero16()};
    static_for<2>([&](auto U) {
        constexpr int u = decltype(U)::value;
        const char* buf = ws.template acquire<0>();
        ws.template begin_c<CB_BWD>();
        mma_tile<16, 0, CB_BWD>(ws, buf, bh, bl, G1[u], G2[u], lane);
    });
#pragma unroll
    for (int s = 0; s < 16; ++s) sh.frag_load(OS_ZB4 * SLOT_BYTES, s, ah[s], al[s]);
    static_for<2>([&](auto U) {
        constexpr int u = decltype(U)::value;
        const char* buf = ws.template acquire<0>();
        if constexpr (u == 0) {
            ws.template begin_c<CB_BWD>();
            mma_tile<16, 0, CB_BWD>(ws, buf, ah, al, G1[u], G2[u], lane);
        } else {   // the next tile's first chunk: the one size that is not a constant
            ws.begin(next_first);
            mma_tile<16, 0, 1>(ws, buf, ah, al, G1[u], G2[u], lane);
        }
    });
    // ---- input map: g_pts = J^T Xb + sum_slots d2(slot) GX(slot) gb[channel of the slot]
    float gp[3] = {0.f, 0.f, 0.f};
    {
        float f[4][8];
        encode_x(p, h, f);
        f32x16 X0 = combine(G1[0], G2[0]), X1 = combine(G1[1], G2[1]);
        const f32x16 c0 = sh.tile_load(OS_GX, 2), c1 = sh.tile_load(OS_GX, 3);
        const f32x16 GX0 = sh.tile_load(OS_GX, 0), GX1 = sh.tile_load(OS_GX, 1);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            X0[i] += c0[i];
            X1[i] += c1[i];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float fr = 1.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float other = other_half(f[c][k], h);
                const float Xv = (c < 2) ? X0[8 * c + k] : X1[k];
                const float Gv = (c < 2) ? GX0[8 * c + k] : GX1[k];
                gp[c] = fmaf(Xv, (h ? -fr : fr) * other, gp[c]);
                gp[c] = fmaf(Gv * gb[c], -(fr * fr) * f[c][k], gp[c]);
                fr *= 2.f;
            }
        }
#pragma unroll
        for (int jj = 0; jj < 6; ++jj) {
            const float fr = (jj & 1) ? 512.f : 256.f;
            const float other = other_half(f[3][jj], h);
            gp[jj >> 1] = fmaf(X1[8 + jj], (h ? -fr : fr) * other, gp[jj >> 1]);
            gp[jj >> 1] = fmaf(GX1[8 + jj] * gb[jj >> 1], -(fr * fr) * f[3][jj], gp[jj >> 1]);
        }
        gp[0] += h ? 0.f : X1[14];
        gp[2] += h ? X1[14] : 0.f;
        gp[1] += h ? 0.f : X1[15];
#pragma unroll
        for (int c = 0; c < 3; ++c) gp[c] = half_sum(gp[c]) * inv_kappa;
    }
    if (valid && h == 0) {
        a.g_pts[3 * n] = gp[0];
        a.g_pts[3 * n + 1] = gp[1];
        a.g_pts[3 * n + 2] = gp[2];
        if (a.g_rays_d != nullptr) {
            if (a.dir_per_sample) {
#pragma unroll
                for (int c = 0; c < 3; ++c) a.g_rays_d[3 * (size_t)n + c] = gdir[c] * inv_kappa;
            } else {
                const int ray = n / a.spr;
#pragma unroll
                for (int c = 0; c < 3; ++c) atomicAdd(a.g_rays_d + 3 * ray + c, gdir[c] * inv_kappa);
            }
        }
    }
}

// MODE 0: sdf only (sampling passes); 1: full evaluation (sdf, d sdf / d p, colour); 2: full evaluation followed by
// its adjoint (hn_field_eval_bwd): the sweeps of oracle/field_bwd.py in the same weight-stream / register-resident
// form, per sample tile, with the tape in the wave's stash.  The fitting step splits mode 2 in two launches so that
// nothing is evaluated twice: 3 = full evaluation that keeps its tape (stash slots per sample TILE, in a buffer the
// caller keeps until the backward pass), 4 = the adjoint alone, from that tape.
template <int MODE>
__global__ __launch_bounds__(256) void k_field2_obj(const Obj2Args a) {
    constexpr bool FULL = MODE >= 1;
    constexpr bool ADJ = MODE >= 2;                    // the forward pass writes the tape
    constexpr bool RUN_FWD = MODE != 4 && MODE != 5;
    constexpr bool RUN_ADJ = MODE == 2 || MODE == 4 || MODE == 5;   // 5: 4 + the per-layer signals of the parameter gradients (Obj2Args::sig)
    constexpr bool PER_TILE = MODE >= 3;               // stash indexed by tile (kept across launches), not by workgroup
    constexpr int N_SLOTS = ADJ ? OBJ2_SLOTS_ADJ : OBJ2_SLOTS;
    constexpr int FIRST_CHUNK = MODE >= 4 ? CB_W4ROWS : CB_L0;   // first chunk of a tile's program
    extern __shared__ __attribute__((aligned(16))) char lds[];
    f16_flush_mode();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably wave-uniform: scalar addressing
    const int j = lane & 31;
    const int h = lane >> 5;
    Stash sh;
    sh.init(FULL ? a.scratch + ((size_t)blockIdx.x * WG_WAVES + wave) * N_SLOTS * SLOT_F4 : nullptr,
            FULL ? N_SLOTS : 0, lane);
    const int n_tiles = (a.n_pts + WG_SAMPLES - 1) / WG_SAMPLES;

    WStream ws;
    ws.init(a.blob, a.blob_bytes, lds, wave, lane);
    ws.dbg_nofetch = (HN_DBG(a) & 4) ? 1 : 0;
    if ((int)blockIdx.x < n_tiles) ws.template fetch_all_c<FIRST_CHUNK>();

    XcdPace xp;   // the workgroups of an XCD meet at every tile start of a long launch (hn_mlp2.h)
    xp.init(a.xsync);
    const int full_rounds = n_tiles / (int)gridDim.x;
    for (int tile = blockIdx.x, it = 0; tile < n_tiles; tile += gridDim.x, ++it) {
        if (xp.on() && it >= 1 && it < full_rounds && it % XCD_PACE_EVERY == 0) xp.meet(it / XCD_PACE_EVERY);
        const bool more = tile + (int)gridDim.x < n_tiles;
        const int n = tile * WG_SAMPLES + wave * 32 + j;
        const bool valid = n < a.n_pts;
        const int nn = valid ? n : a.n_pts - 1;
        const float p[3] = {a.pts[3 * nn], a.pts[3 * nn + 1], a.pts[3 * nn + 2]};
        if constexpr (PER_TILE) sh.init(a.scratch + ((size_t)tile * WG_WAVES + wave) * N_SLOTS * SLOT_F4, N_SLOTS, lane);
        h8 ah[16], al[16], bh[16], bl[16];   // ping-pong activation fragments
        float g[3] = {0.f, 0.f, 0.f}, rgb[3] = {0.f, 0.f, 0.f};
        float sdf = 0.f;
        const int ray = nn / a.spr;
        const float d[3] = {FULL ? a.rays_d[3 * ray] : 0.f, FULL ? a.rays_d[3 * ray + 1] : 0.f, FULL ? a.rays_d[3 * ray + 2] : 0.f};
        if constexpr (!RUN_FWD) {   // the adjoint alone: the evaluation's outputs come from the forward launch
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                g[c] = a.grad[3 * nn + c];
                rgb[c] = a.rgb[3 * nn + c];
            }
        } else {

        // X space fragments (lin0, lin4 skip, colour lin0).  The full kernel parks them in the stash
        // between uses (40 registers over ~150 chunks); the sdf-only kernel has room to keep them.
        h8 xh[4], xl[4];
        {
            float f[4][8];
            encode_x(p, h, f);
#pragma unroll
            for (int s = 0; s < 4; ++s) split8(f[s], xh[s], xl[s]);
            if constexpr (FULL) {
#pragma unroll
                for (int s = 0; s < 4; ++s) sh.frag_store(OS_X * SLOT_BYTES, s, xh[s], xl[s]);
            }
        }
        struct Act {
            f32x16 v;
        };
        auto no_pre = [](auto, const char*) { return NoData{}; };
        auto no_store = [](auto, const auto&) {};
        struct Frags {
            h8 hi[2], lo[2];
        };
        // fragments of the finished tile -> the next layer's input registers
        auto to_regs = [&](h8(&oh)[16], h8(&ol)[16]) {
            return [&oh, &ol](auto T, EpiState& st, const auto&) {
                constexpr int t = decltype(T)::value;
                asm volatile("" : "+v"(st.hi[0]), "+v"(st.lo[0]), "+v"(st.hi[1]), "+v"(st.lo[1]));
                oh[2 * t] = st.hi[0];
                ol[2 * t] = st.lo[0];
                oh[2 * t + 1] = st.hi[1];
                ol[2 * t + 1] = st.lo[1];
#if HN_PARK_AGPR
                asm volatile("" : "+a"(oh[2 * t]), "+a"(ol[2 * t]), "+a"(oh[2 * t + 1]), "+a"(ol[2 * t + 1]));   // as hn_field2_hand.hip: park
#endif
                return NoData{};
            };
        };
        // ... and, in the full kernel, the fp32 activation to the stash for the reverse sweep
        auto to_regs_keep = [&](h8(&oh)[16], h8(&ol)[16], int stash_slot) {
            return [&oh, &ol, stash_slot, &sh, &a](auto T, EpiState& st, const auto&) {
                constexpr int t = decltype(T)::value;
                (void)a;
                asm volatile("" : "+v"(st.hi[0]), "+v"(st.lo[0]), "+v"(st.hi[1]), "+v"(st.lo[1]));
                oh[2 * t] = st.hi[0];
                ol[2 * t] = st.lo[0];
                oh[2 * t + 1] = st.hi[1];
                ol[2 * t + 1] = st.lo[1];
#if HN_PARK_AGPR
                asm volatile("" : "+a"(oh[2 * t]), "+a"(ol[2 * t]), "+a"(oh[2 * t + 1]), "+a"(ol[2 * t + 1]));   // as hn_field2_hand.hip: park
#endif
                if (FULL && !(HN_DBG(a) & 1)) sh.tile_store(stash_slot, t, st.vec());
                return NoData{};
            };
        };
        auto stash_frags = [&](int stash_slot) {
            return [stash_slot, &sh](auto T, const Frags& f) {
                constexpr int t = decltype(T)::value;
                sh.frag_store(stash_slot * SLOT_BYTES, 2 * t, f.hi[0], f.lo[0]);
                sh.frag_store(stash_slot * SLOT_BYTES, 2 * t + 1, f.hi[1], f.lo[1]);
            };
        };

        // ---- lin0: X -> a1 (2 chunks of 4 tiles x 4 k-steps)
        {
            h8 x16h[16], x16l[16];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                x16h[s] = xh[s];
                x16l[s] = xl[s];
            }
            run_layer_c<8, 4, 4, true, true, CB_L0, CB_HID>(ws, x16h, x16l, lane, h, no_pre, PhSoftplus{}, to_regs_keep(ah, al, OS_A1 + 0), no_store);
        }
        run_layer_c<8, 16, 1, true, true, CB_HID, CB_HID>(ws, ah, al, lane, h, no_pre, PhSoftplus{}, to_regs_keep(bh, bl, OS_A1 + 1), no_store);   // lin1
        run_layer_c<8, 16, 1, true, true, CB_HID, CB_HID>(ws, bh, bl, lane, h, no_pre, PhSoftplus{}, to_regs_keep(ah, al, OS_A1 + 2), no_store);   // lin2
        // ---- lin3: 193 outputs = 7 tiles (tile 6 holds neuron 192 in row 0)
        float a4_192 = 0.f;
        run_layer_c<7, 16, 1, true, true, CB_HID, CB_HID>(ws, ah, al, lane, h, no_pre, PhSoftplus{},
                                        [&](auto T, EpiState& st, const auto&) {
                                            constexpr int t = decltype(T)::value;
                                            asm volatile("" : "+v"(st.hi[0]), "+v"(st.lo[0]), "+v"(st.hi[1]), "+v"(st.lo[1]));
                                            if constexpr (t == 6) {
                                                // rows 193..223 are padding (zero weights and bias give softplus(0)): drop them
#pragma unroll
                                                for (int i = 0; i < 16; ++i) st.v[i] = (i == 0 && h == 0) ? st.v[i] : 0.f;
                                                a4_192 = st.v[0];
                                            } else {
                                                bh[2 * t] = st.hi[0];
                                                bl[2 * t] = st.lo[0];
                                                bh[2 * t + 1] = st.hi[1];
                                                bl[2 * t + 1] = st.lo[1];
                                            }
                                            if (FULL && !(HN_DBG(a) & 1)) sh.tile_store(OS_A1 + 3, t, st.vec());
                                            return NoData{};
                                        },
                                        no_store);
        // ---- lin4: [a4 (192 via k-steps 0..11) | X with a4[192] in its pad slot] / sqrt2
        {
            const float v192 = other_half(a4_192, h);   // half 1 receives half 0's value
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if constexpr (FULL) {
                    sh.frag_load(OS_X * SLOT_BYTES, s, bh[12 + s], bl[12 + s]);
                } else {
                    bh[12 + s] = xh[s];
                    bl[12 + s] = xl[s];
                }
            }
            // the pad slot (k-step 3, half 1, element 7) carries a4[192]
            const _Float16 vh = hi_part(v192);
            const _Float16 vl = (_Float16)((v192 - (float)vh) * LO_SCALE);
            bh[15][7] = h ? vh : bh[15][7];
            bl[15][7] = h ? vl : bl[15][7];
        }
        run_layer_c<8, 16, 1, true, true, CB_HID, CB_HID>(ws, bh, bl, lane, h, no_pre, PhSoftplus{}, to_regs_keep(ah, al, OS_A1 + 4), no_store);   // lin4
        run_layer_c<8, 16, 1, true, true, CB_HID, CB_HID>(ws, ah, al, lane, h, no_pre, PhSoftplus{}, to_regs_keep(bh, bl, OS_A1 + 5), no_store);   // lin5
        run_layer_c<8, 16, 1, true, true, CB_HID, CB_HID>(ws, bh, bl, lane, h, no_pre, PhSoftplus{}, to_regs_keep(ah, al, OS_A1 + 6), no_store);   // lin6
        // ---- lin7 -> a8; sdf = W8[0,:] a8 + b8; seed of the reverse sweep dz7 = sigma'(z7) W8[0,:] / scale
        float sdf_acc = 0.f;
        auto lin7 = [&](auto NA_) {   // NA_: the size of the chunk that follows the layer, a constant
        run_layer_c<8, 16, 1, true, true, CB_HID, decltype(NA_)::value>(
            ws, ah, al, lane, h,
            [&](auto, const char* tail) { return Act{tail_tile(tail, 1, h)}; }, PhSoftplus{},
            [&](auto T, EpiState& st, const Act& w8) {
                constexpr int t = decltype(T)::value;
                asm volatile("" : "+v"(st.hi[0]), "+v"(st.lo[0]), "+v"(st.hi[1]), "+v"(st.lo[1]));
                f32x16 dz;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    sdf_acc = fmaf(w8.v[i], st.v[i], sdf_acc);
                    dz[i] = dsoftplus_from_act(st.v[i]) * w8.v[i] * (a.inv_scale * BWD_SCALE);
                }
                if (FULL) {
                    bh[2 * t] = st.hi[0];   // a8 feeds lin8
                    bl[2 * t] = st.lo[0];
                    bh[2 * t + 1] = st.hi[1];
                    bl[2 * t + 1] = st.lo[1];
                    Frags f;
                    split_tile(dz, f.hi[0], f.lo[0], f.hi[1], f.lo[1]);
                    stash_frags(OS_DZ7)(T, f);
                    if constexpr (ADJ) {
                        sh.tile_store(OS_A8, t, st.vec());
                        sh.tile_store(OS_DZ + 7, t, dz);
                    }
                }
                return NoData{};
            },
            no_store);
        };
        if constexpr (FULL) {
            lin7(std::integral_constant<int, CB_HID>{});
        } else {
            if (more)
                lin7(std::integral_constant<int, CB_L0>{});
            else
                lin7(std::integral_constant<int, 0>{});
        }
        sdf = (half_sum(sdf_acc) + a.b8[0]) * a.inv_scale;
        if (!FULL) {
            if (valid && h == 0) a.sdf[n] = sdf;
            continue;
        }
        if (HN_DBG(a) & 8) {   // bisecting aid: stop after the forward pass (single tile per workgroup only)
            if (valid && h == 0) a.sdf[n] = sdf;
            return;
        }

        // ---- lin8 rows 1..256: the feature vector (no activation) -> stash as fragments for colour lin0
        run_layer_c<8, 16, 1, true, true, CB_HID, CB_BWD>(
            ws, bh, bl, lane, h, no_pre, PhIdentity{},
            [&](auto T, EpiState& st, const auto&) {
                constexpr int t = decltype(T)::value;
                if (a.feat != nullptr && valid) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) a.feat[(size_t)n * H + 32 * t + tile_row(i, h)] = st.v[i];
                }
                stash_frags(OS_FVEC)(T, Frags{{st.hi[0], st.hi[1]}, {st.lo[0], st.lo[1]}});
                return NoData{};
            },
            no_store);

        // ---- reverse sweep: dz_{l-1} = sigma'(z_{l-1}) * (W_l^T dz_l); sigma' from the stashed activation a_l
        auto act_of = [&](int act_slot) {
            return [&sh, act_slot, &a](auto T, const char*) {
                (void)a;
                if (HN_DBG(a) & 2) return Act{zero16()};
                return Act{sh.tile_load(act_slot, decltype(T)::value)};
            };
        };
#pragma unroll
        for (int s = 0; s < 16; ++s) sh.frag_load(OS_DZ7 * SLOT_BYTES, s, ah[s], al[s]);
        // (adjoint mode: every dz_l also goes to the stash as an fp32 tile, slot OS_DZ + l)
        auto to_regs_dz = [&](h8(&oh)[16], h8(&ol)[16], int dz_slot) {
            return [&oh, &ol, dz_slot, &sh](auto T, EpiState& st, const auto&) {
                constexpr int t = decltype(T)::value;
                asm volatile("" : "+v"(st.hi[0]), "+v"(st.lo[0]), "+v"(st.hi[1]), "+v"(st.lo[1]));
                oh[2 * t] = st.hi[0];
                ol[2 * t] = st.lo[0];
                oh[2 * t + 1] = st.hi[1];
                ol[2 * t + 1] = st.lo[1];
#if HN_PARK_AGPR
                asm volatile("" : "+a"(oh[2 * t]), "+a"(ol[2 * t]), "+a"(oh[2 * t + 1]), "+a"(ol[2 * t + 1]));
#endif
                if constexpr (ADJ) sh.tile_store(dz_slot, t, st.vec());
                return NoData{};
            };
        };
        run_layer_c<8, 16, 1, false, true, CB_BWD, CB_BWD>(ws, ah, al, lane, h, act_of(OS_A1 + 6), PhDsig{}, to_regs_dz(bh, bl, OS_DZ + 6), no_store);   // W7^T -> dz6
        run_layer_c<8, 16, 1, false, true, CB_BWD, CB_BWD>(ws, bh, bl, lane, h, act_of(OS_A1 + 5), PhDsig{}, to_regs_dz(ah, al, OS_DZ + 5), no_store);   // W6^T -> dz5
        run_layer_c<8, 16, 1, false, true, CB_BWD, CB_BWD>(ws, ah, al, lane, h, act_of(OS_A1 + 4), PhDsig{},                          // W5^T -> dz4 (kept)
                                         [&](auto T, EpiState& st, const auto&) {
                                             constexpr int t = decltype(T)::value;
                                             asm volatile("" : "+v"(st.hi[0]), "+v"(st.lo[0]), "+v"(st.hi[1]), "+v"(st.lo[1]));
                                             bh[2 * t] = st.hi[0];
                                             bl[2 * t] = st.lo[0];
                                             bh[2 * t + 1] = st.hi[1];
                                             bl[2 * t + 1] = st.lo[1];
                                             stash_frags(OS_DZ4)(T, Frags{{st.hi[0], st.hi[1]}, {st.lo[0], st.lo[1]}});
                                             if constexpr (ADJ) sh.tile_store(OS_DZ + 4, t, st.vec());
                                             return NoData{};
                                         },
                                         no_store);
        // W4[:, :193]^T: dz4 -> dz3 (193 rows = 7 tiles; a4's padding rows were stashed as 0 => sigma' = 0)
        run_layer_c<7, 16, 1, false, true, CB_BWD, CB_BWD3>(ws, bh, bl, lane, h, act_of(OS_A1 + 3), PhDsig{},
                                         [&](auto T, EpiState& st, const auto&) {
                                             constexpr int t = decltype(T)::value;
                                             asm volatile("" : "+v"(st.hi[0]), "+v"(st.lo[0]), "+v"(st.hi[1]), "+v"(st.lo[1]));
                                             if constexpr (t < 6) {
                                                 ah[2 * t] = st.hi[0];
                                                 al[2 * t] = st.lo[0];
                                                 ah[2 * t + 1] = st.hi[1];
                                                 al[2 * t + 1] = st.lo[1];
                                             } else {
                                                 ah[12] = st.hi[0];   // only k-step 12 exists (neuron 192); 13 is padding
                                                 al[12] = st.lo[0];
                                             }
                                             if constexpr (ADJ) sh.tile_store(OS_DZ + 3, t, st.vec());
                                             return NoData{};
                                         },
                                         no_store);
        run_layer_c<8, 13, 1, false, true, CB_BWD3, CB_BWD>(ws, ah, al, lane, h, act_of(OS_A1 + 2), PhDsig{}, to_regs_dz(bh, bl, OS_DZ + 2), no_store);   // W3^T -> dz2
        run_layer_c<8, 16, 1, false, true, CB_BWD, CB_BWD>(ws, bh, bl, lane, h, act_of(OS_A1 + 1), PhDsig{}, to_regs_dz(ah, al, OS_DZ + 1), no_store);    // W2^T -> dz1
        run_layer_c<8, 16, 1, false, true, CB_BWD, CB_BWD>(ws, ah, al, lane, h, act_of(OS_A1 + 0), PhDsig{}, to_regs_dz(bh, bl, OS_DZ + 0), no_store);    // W1^T -> dz0
        // d sdf / d X-space = W0^T dz0 + W4[:, 193:]^T dz4   (64 rows = 2 tiles; row <-> k-slot of the same lane)
        f32x16 G1[2] = {zero16(), zero16()}, G2[2] = {zero16(), zero16()};
        static_for<2>([&](auto U) {
            constexpr int u = decltype(U)::value;
            const char* buf = ws.template acquire<0>();
            ws.template begin_c<CB_BWD>();
            mma_tile<16, 0, CB_BWD>(ws, buf, bh, bl, G1[u], G2[u], lane);
        });
#pragma unroll
        for (int s = 0; s < 16; ++s) sh.frag_load(OS_DZ4 * SLOT_BYTES, s, ah[s], al[s]);
        static_for<2>([&](auto U) {
            constexpr int u = decltype(U)::value;
            const char* buf = ws.template acquire<0>();
            constexpr int nbytes = u == 0 ? CB_BWD : CB_C0A;
            ws.template begin_c<nbytes>();
            mma_tile<16, 0, nbytes>(ws, buf, ah, al, G1[u], G2[u], lane);
        });
        // ---- Jacobian of the encoding (in-lane: G row of tile u, register 8(s&1)+j <-> k-slot (s = 2u + .., h, j))
        {
            float f[4][8];
            encode_x(p, h, f);
            f32x16 G0 = combine(G1[0], G2[0]), Gb = combine(G1[1], G2[1]);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                G0[i] *= BWD_INV;
                Gb[i] *= BWD_INV;
            }
            if constexpr (ADJ) {   // d sdf / d X, for the second-order term of the input map
                sh.tile_store(OS_GX, 0, G0);
                sh.tile_store(OS_GX, 1, Gb);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float fr = 1.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float other = other_half(f[c][k], h);   // the conjugate function of the same angle
                    const float Gv = (c < 2) ? G0[8 * c + k] : Gb[k];
                    g[c] = fmaf(Gv, (h ? -fr : fr) * other, g[c]);
                    fr *= 2.f;
                }
            }
#pragma unroll
            for (int jj = 0; jj < 6; ++jj) {
                const float fr = (jj & 1) ? 512.f : 256.f;
                const float other = other_half(f[3][jj], h);
                g[jj >> 1] = fmaf(Gb[8 + jj], (h ? -fr : fr) * other, g[jj >> 1]);
            }
            g[0] += h ? 0.f : Gb[14];
            g[2] += h ? Gb[14] : 0.f;
            g[1] += h ? 0.f : Gb[15];
            g[0] = half_sum(g[0]);
            g[1] = half_sum(g[1]);
            g[2] = half_sum(g[2]);
        }
        // ---- colour lin0: [enc(p) | enc(d) | feature vector | enc(g)] -> relu
        h8 mh[8], ml[8];   // the 8 k-steps of chunk B: X (4), enc(d) (2), enc(g) (2)
        {
#pragma unroll
            for (int s = 0; s < 4; ++s) sh.frag_load(OS_X * SLOT_BYTES, s, mh[s], ml[s]);
            float fd[2][8], fg[2][8];
            encode_v4(d, h, fd);
            encode_v4(g, h, fg);
            split8(fd[0], mh[4], ml[4]);
            split8(fd[1], mh[5], ml[5]);
            split8(fg[0], mh[6], ml[6]);
            split8(fg[1], mh[7], ml[7]);
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) sh.frag_load(OS_FVEC * SLOT_BYTES, s, ah[s], al[s]);
        {
            // two chunks per tile (16 feature-vector k-steps, then 8 encoding k-steps + bias); the epilogue of
            // tile t-1 rides on chunk A's MFMAs of tile t
            f32x16 c1[2], c2[2];
            EpiState st;
            NoData nd;
            PhRelu relu;
            auto put = [&](auto T) {
                constexpr int t = decltype(T)::value;
                asm volatile("" : "+v"(st.hi[0]), "+v"(st.lo[0]), "+v"(st.hi[1]), "+v"(st.lo[1]));
                bh[2 * t] = st.hi[0];
                bl[2 * t] = st.lo[0];
                bh[2 * t + 1] = st.hi[1];
                bl[2 * t + 1] = st.lo[1];
                if constexpr (ADJ) sh.tile_store(OS_C + 0, t, st.vec());
            };
            static_for<8>([&](auto T) {
                constexpr int t = decltype(T)::value;
                const char* bufa = ws.template acquire<0>();
                ws.template begin_c<CB_C0B>();
                arm(st);
                if constexpr (t > 0) {
                    st.c1 = c1[(t - 1) & 1];
                    st.c2 = c2[(t - 1) & 1];
                }
                c1[t & 1] = zero16();
                c2[t & 1] = zero16();
                if constexpr (t > 0) {
                    Epi<true, PhRelu, NoData> epi{st, relu, nd};
                    mma_tile<16, 0, CB_C0B>(ws, bufa, ah, al, c1[t & 1], c2[t & 1], lane, epi);
                    split_finish<true>(st);
                    put(std::integral_constant<int, t - 1>{});
                } else {
                    mma_tile<16, 0, CB_C0B>(ws, bufa, ah, al, c1[t & 1], c2[t & 1], lane);
                }
                const char* bufb = ws.template acquire<0>();
                constexpr int nbytes = t + 1 < 8 ? CB_C0A : CB_HID;
                ws.template begin_c<nbytes>();
                const f32x16 bias = tail_tile(bufb + 8 * KS_BYTES, 0, h);
                mma_tile<8, 0, nbytes>(ws, bufb, mh, ml, c1[t & 1], c2[t & 1], lane);
#pragma unroll
                for (int i = 0; i < 16; ++i) c1[t & 1][i] += bias[i];
            });
            arm(st);
            st.c1 = c1[1];
            st.c2 = c2[1];
            Epi<true, PhRelu, NoData> epi{st, relu, nd};
            epi.run_all();
            split_finish<true>(st);
            put(std::integral_constant<int, 7>{});
        }
        // (adjoint mode: the colour activations go to the stash for the relu masks)
        auto to_regs_c = [&](h8(&oh)[16], h8(&ol)[16], int c_slot) {
            return [&oh, &ol, c_slot, &sh](auto T, EpiState& st, const auto&) {
                constexpr int t = decltype(T)::value;
                asm volatile("" : "+v"(st.hi[0]), "+v"(st.lo[0]), "+v"(st.hi[1]), "+v"(st.lo[1]));
                oh[2 * t] = st.hi[0];
                ol[2 * t] = st.lo[0];
                oh[2 * t + 1] = st.hi[1];
                ol[2 * t + 1] = st.lo[1];
#if HN_PARK_AGPR
                asm volatile("" : "+a"(oh[2 * t]), "+a"(ol[2 * t]), "+a"(oh[2 * t + 1]), "+a"(ol[2 * t + 1]));
#endif
                if constexpr (ADJ) sh.tile_store(c_slot, t, st.vec());
                return NoData{};
            };
        };
        run_layer_c<8, 16, 1, true, true, CB_HID, CB_HID>(ws, bh, bl, lane, h, no_pre, PhRelu{}, to_regs_c(ah, al, OS_C + 1), no_store);   // colour lin1
        run_layer_c<8, 16, 1, true, true, CB_HID, CB_HID>(ws, ah, al, lane, h, no_pre, PhRelu{}, to_regs_c(bh, bl, OS_C + 2), no_store);   // colour lin2
        struct W3 {
            f32x16 w[3];
        };
        auto col3 = [&](auto NA_) {
        run_layer_c<8, 16, 1, true, false, CB_HID, decltype(NA_)::value>(   // colour lin3 + the 3 rows of lin4 (tail slots 1..3)
            ws, bh, bl, lane, h,
            [&](auto, const char* tail) { return W3{{tail_tile(tail, 1, h), tail_tile(tail, 2, h), tail_tile(tail, 3, h)}}; },
            PhRelu{},
            [&](auto T, EpiState& st, const W3& w) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int i = 0; i < 16; ++i) rgb[c] = fmaf(w.w[c][i], st.v[i], rgb[c]);
                asm volatile("" : "+v"(rgb[0]), "+v"(rgb[1]), "+v"(rgb[2]));
                if constexpr (ADJ) sh.tile_store(OS_C + 3, decltype(T)::value, st.vec());
                return NoData{};
            },
            no_store);
        };
        if constexpr (MODE == 2) {
            col3(std::integral_constant<int, CB_W4ROWS>{});
        } else {
            if (more)
                col3(std::integral_constant<int, CB_L0>{});
            else
                col3(std::integral_constant<int, 0>{});
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[c] = sigmoid_fast(half_sum(rgb[c]) + a.c_blast[c]);
        }   // RUN_FWD
        if constexpr (RUN_ADJ) {
            obj_adjoint<MODE == 5>(a, ws, sh, lane, h, n, nn, valid, more ? FIRST_CHUNK : 0, p, d, g, rgb, ah, al, bh, bl);
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

#ifndef HN_OBJ_QUAD_TU   // hn_field2_obj_q.hip includes this file for its device helpers only
constexpr size_t OBJ2_LDS = 2 * CHUNK_MAX;

int launch_field2_obj_q(const Obj2Args& a, int n_blocks, int n_cus, hipStream_t stream);   // hn_field2_obj_q.hip
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
static int obj2_grid(int n_pts, int n_cus) {
    const int n_tiles = (n_pts + WG_SAMPLES - 1) / WG_SAMPLES;
    return n_tiles < n_cus ? n_tiles : n_cus;
}

size_t field2_obj_workspace_bytes(int n_pts, int n_cus) {
    return (size_t)obj2_grid(n_pts, n_cus) * WG_WAVES * OBJ2_SLOTS * SLOT_F4 * sizeof(float4) + 256;   // + the XCD pacing counters
}

// bytes of the tape a taped full evaluation leaves for the adjoint launch: the adjoint's stash slots per sample TILE
size_t field2_obj_tape_bytes(int n_pts) {
    const size_t n_tiles = ((size_t)(n_pts > 0 ? n_pts : 0) + WG_SAMPLES - 1) / WG_SAMPLES;
    return n_tiles * WG_WAVES * OBJ2_SLOTS_ADJ * SLOT_F4 * sizeof(float4);
}

// tape != NULL (full evaluation only): MODE 3, the evaluation keeps its tape there (field2_obj_tape_bytes) instead of
// using the workspace
int launch_field2_obj(const hn_field* f, const float* pts, const float* rays_d, int n_pts, int spr, float* sdf,
                      float* grad, float* rgb, float* feat, void* workspace, size_t workspace_bytes, bool full,
                      hipStream_t stream, void* tape = nullptr, size_t tape_bytes = 0) {
    if (n_pts <= 0) return HN_OK;
    Obj2Args a{};
    a.pts = pts;
    a.rays_d = rays_d;
    a.n_pts = n_pts;
    a.spr = spr > 0 ? spr : 1;
    a.inv_scale = 1.f / f->scale;
    a.blob = reinterpret_cast<const char*>(full ? f->v2_full : f->v2_sdf);
    a.blob_bytes = full ? f->v2_full_bytes : f->v2_sdf_bytes;
    if (a.blob == nullptr) {
        set_error("field was not created with HN_PREC_F16X3");
        return HN_EINVAL;
    }
    a.b8 = f->raw_sdf_b[8];
    a.c_blast = f->raw_col_b[4];
    a.sdf = sdf;
    a.grad = grad;
    a.rgb = rgb;
    a.feat = feat;
    a.scratch = reinterpret_cast<float4*>(workspace);
    a.dbg = 0;
#ifdef HN_DEBUG_HOOKS
    {
        const char* e = getenv("HN_DBG");
        a.dbg = e ? atoi(e) : 0;
    }
#endif
    int n_cus = device_cus();
    if (n_cus <= 0) n_cus = 256;
    const int grid = obj2_grid(n_pts, n_cus);
    if (full && tape != nullptr) {
        if (tape_bytes < field2_obj_tape_bytes(n_pts)) {
            set_error("field tape too small: %zu < %zu", tape_bytes, field2_obj_tape_bytes(n_pts));
            return HN_ENOMEM;
        }
        a.scratch = reinterpret_cast<float4*>(tape);
        static std::atomic<uint64_t> lds_tape{0};
        HN_TRY_RC(ensure_dynamic_lds(reinterpret_cast<const void*>(k_field2_obj<3>), (int)OBJ2_LDS, &lds_tape));
        hipLaunchKernelGGL(k_field2_obj<3>, dim3(taped_grid(n_pts, n_cus)), dim3(256), OBJ2_LDS, stream, a);
        HN_LAUNCH_CHECK();
        return HN_OK;
    }
    if (full) {
        const size_t need = (size_t)grid * WG_WAVES * OBJ2_SLOTS * SLOT_F4 * sizeof(float4);
        if (workspace == nullptr || workspace_bytes < need) {
            set_error("field workspace too small: %zu < %zu", workspace_bytes, need);
            return HN_ENOMEM;
        }
        if (HN_XCD_PACING && (n_pts + WG_SAMPLES - 1) / WG_SAMPLES >= XCD_PACE_MIN_ROUNDS * grid && workspace_bytes >= need + 64) {
            a.xsync = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(workspace) + need);
            HN_CHECK_HIP(hipMemsetAsync(a.xsync, 0, 64, stream));
        if (const int ph = pace_phantom_members()) HN_CHECK_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(a.xsync), ph, 8, stream));
        }
    }
    // an sdf-only launch too small to fill the chip: the latency form (hn_field2_obj_q.hip; bit-identical results)
    {
        const int n_blocks = (n_pts + 31) / 32;
        const int qmax = quad_max_blocks_override();
        if (!full && n_blocks <= (qmax >= 0 ? qmax : 2 * n_cus)) return launch_field2_obj_q(a, n_blocks, n_cus, stream);
    }
    static std::atomic<uint64_t> lds_full{0}, lds_sdf{0};   // devices on which the LDS size attribute is set
    HN_TRY_RC(ensure_dynamic_lds(reinterpret_cast<const void*>(k_field2_obj<1>), (int)OBJ2_LDS, &lds_full));
    HN_TRY_RC(ensure_dynamic_lds(reinterpret_cast<const void*>(k_field2_obj<0>), (int)OBJ2_LDS, &lds_sdf));
    if (full)
        hipLaunchKernelGGL(k_field2_obj<1>, dim3(grid), dim3(256), OBJ2_LDS, stream, a);
    else
        hipLaunchKernelGGL(k_field2_obj<0>, dim3(grid), dim3(256), OBJ2_LDS, stream, a);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

int field2_obj_signal_arrays() { return OSG_COUNT; }
size_t field2_obj_adj_workspace_bytes(int n_pts, int n_cus) {
    return (size_t)obj2_grid(n_pts, n_cus) * WG_WAVES * OBJ2_SLOTS_ADJ * SLOT_F4 * sizeof(float4);
}

// hn_field_eval_bwd for an HN_PREC_F16X3 object field: one persistent launch (evaluation + adjoint per sample tile).
// tape != NULL: the adjoint alone (MODE 4) from the tape of a taped evaluation of the same points, whose outputs
// `grad`, `rgb` are passed back in.
int launch_field2_obj_adj(const hn_field* f, const float* pts, const float* rays_d, int n_pts, int spr, const float* g_sdf,
                          const float* g_grad, const float* g_rgb, float* g_pts, float* g_rays_d, void* workspace,
                          size_t workspace_bytes, hipStream_t stream, const void* tape = nullptr, const float* grad = nullptr,
                          const float* rgb = nullptr, float* sig = nullptr, size_t sig_pitch = 0, float* gb_out = nullptr) {
    if (n_pts <= 0) return HN_OK;
    HN_REQUIRE((tape != nullptr ? f->v2_adjonly : f->v2_adj) != nullptr, "field has no adjoint program");
    HN_REQUIRE(tape == nullptr || (grad != nullptr && rgb != nullptr), "the adjoint from a tape needs the evaluation's grad / rgb");
    HN_REQUIRE(sig == nullptr || (tape != nullptr && gb_out != nullptr && sig_pitch >= (size_t)n_pts * 256), "the signal arrays belong to the adjoint from a tape");
    Obj2Args a{};
    a.sig = sig;
    a.sig_pitch = sig_pitch;
    a.gb_out = gb_out;
    a.pts = pts;
    a.rays_d = rays_d;
    a.n_pts = n_pts;
    a.spr = spr > 0 ? spr : 1;
    a.inv_scale = 1.f / f->scale;
    a.blob = reinterpret_cast<const char*>(f->v2_adj);
    a.blob_bytes = f->v2_adj_bytes;
    a.b8 = f->raw_sdf_b[8];
    a.c_blast = f->raw_col_b[4];
    a.scratch = reinterpret_cast<float4*>(workspace);
    a.dbg = 0;
    a.g_sdf = g_sdf;
    a.g_grad = g_grad;
    a.g_rgb = g_rgb;
    a.g_pts = g_pts;
    a.g_rays_d = g_rays_d;
    int n_cus = device_cus();
    if (n_cus <= 0) n_cus = 256;
    const int grid = obj2_grid(n_pts, n_cus);
    a.dir_per_sample = launch_dir_per_sample() ? 1 : 0;
    if (g_rays_d != nullptr && !a.dir_per_sample) HN_CHECK_HIP(hipMemsetAsync(g_rays_d, 0, (size_t)(n_pts / a.spr) * 3 * sizeof(float), stream));
    if (tape != nullptr) {
        a.blob = reinterpret_cast<const char*>(f->v2_adjonly);
        a.blob_bytes = f->v2_adjonly_bytes;
        a.scratch = reinterpret_cast<float4*>(const_cast<void*>(tape));
        a.grad = const_cast<float*>(grad);   // read only in this mode
        a.rgb = const_cast<float*>(rgb);
        if (sig != nullptr) {   // MODE 5: the same adjoint, leaving the per-layer signals of the parameter gradients
            static std::atomic<uint64_t> lds_sig{0};
            HN_TRY_RC(ensure_dynamic_lds(reinterpret_cast<const void*>(k_field2_obj<5>), (int)OBJ2_LDS, &lds_sig));
            hipLaunchKernelGGL(k_field2_obj<5>, dim3(taped_grid(n_pts, n_cus)), dim3(256), OBJ2_LDS, stream, a);
            HN_LAUNCH_CHECK();
            return HN_OK;
        }
        static std::atomic<uint64_t> lds_adjonly{0};
        HN_TRY_RC(ensure_dynamic_lds(reinterpret_cast<const void*>(k_field2_obj<4>), (int)OBJ2_LDS, &lds_adjonly));
        hipLaunchKernelGGL(k_field2_obj<4>, dim3(taped_grid(n_pts, n_cus)), dim3(256), OBJ2_LDS, stream, a);
        HN_LAUNCH_CHECK();
        return HN_OK;
    }
    const size_t need = (size_t)grid * WG_WAVES * OBJ2_SLOTS_ADJ * SLOT_F4 * sizeof(float4);
    if (workspace == nullptr || workspace_bytes < need) {
        set_error("adjoint workspace too small: %zu < %zu", workspace_bytes, need);
        return HN_ENOMEM;
    }
    static std::atomic<uint64_t> lds_adj{0};
    HN_TRY_RC(ensure_dynamic_lds(reinterpret_cast<const void*>(k_field2_obj<2>), (int)OBJ2_LDS, &lds_adj));
    hipLaunchKernelGGL(k_field2_obj<2>, dim3(grid), dim3(256), OBJ2_LDS, stream, a);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

#endif   // HN_OBJ_QUAD_TU
}  // namespace v2
}  // namespace hn

#if defined(HN_TS) && !defined(HN_OBJ_QUAD_TU)
// timing builds (tools/ts_report_obj.py): workgroup 0's per-chunk stamps of the last object-field launch
extern "C" int hn_debug_ts_obj(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(hn::v2::g_hn_ts), sizeof(unsigned long long) * n);
}
#endif
