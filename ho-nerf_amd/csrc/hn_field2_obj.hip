// v2 object field (f16x3 MFMA, LDS-streamed weights): SDFNetwork_OBJ forward, analytic
// d sdf / d p (reverse sweep) and RenderingNetwork_OBJ, fused; 4 waves x 32 samples per
// workgroup, one workgroup per CU, activations in registers (hn_mlp2.h).
//
// Reference: utils/fields.py:316-347 (sdf net, .gradient), :387-405 (colour net), called from
// utils/renderer.py:130-135 / 380-385.  The chunk order below is the contract with
// hn_pack2.hip (build_obj_stream).
#include <stdlib.h>

#include "hn_mlp2.h"

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
    float b8;
    float c_blast[3];
    float* sdf;
    float* grad;
    float* rgb;
    float* feat;           // optional [n,256]
    float4* scratch;       // per-wave stash slots (FULL only)
    int dbg;               // timing experiments only (HN_DBG): 1 = no stash stores, 2 = no stash loads
};

// stash slots of one wave (32 KiB each)
enum { OS_A1 = 0 /* a1..a7 -> 0..6 */, OS_DZ7 = 7, OS_FVEC = 8, OS_DZ4 = 9, OBJ2_SLOTS = 10 };

constexpr int CB_HID = chunk_bytes(1, 16, true);     // hidden layer tile: 16 k-steps + tail
constexpr int CB_L0 = chunk_bytes(4, 4, true);       // lin0: 4 tiles x 4 k-steps + tail
constexpr int CB_BWD = chunk_bytes(1, 16, false);    // transposed hidden tile
constexpr int CB_BWD3 = chunk_bytes(1, 13, false);   // W3^T: 193 outputs = 13 k-steps
constexpr int CB_C0A = chunk_bytes(1, 16, false);    // colour lin0, feature-vector columns
constexpr int CB_C0B = chunk_bytes(1, 8, true);      // colour lin0, enc(p) | enc(d) | enc(g) columns + bias

// sin/cos(2^k x) of one lane half: half 0 keeps the sines, half 1 the cosines
__device__ __forceinline__ float sc_half(float ang, int h) {
    float s, c;
    sincosf(ang, &s, &c);
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

template <bool FULL>
__global__ __launch_bounds__(256) void k_field2_obj(const Obj2Args a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably wave-uniform: scalar addressing
    const int j = lane & 31;
    const int h = lane >> 5;
    float4* wslot = a.scratch + ((size_t)blockIdx.x * WG_WAVES + wave) * OBJ2_SLOTS * SLOT_F4;
    auto slot = [&](int i) { return wslot + (size_t)i * SLOT_F4; };
    const int n_tiles = (a.n_pts + WG_SAMPLES - 1) / WG_SAMPLES;

    WStream ws;
    ws.g = a.blob;
    ws.begin = a.blob;
    ws.end = a.blob + a.blob_bytes;
    ws.lds = lds;
    ws.phase = 0;
    ws.wave = wave;
    ws.lane = lane;
    if ((int)blockIdx.x < n_tiles) ws.fetch(CB_L0);

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const bool more = tile + (int)gridDim.x < n_tiles;
        wslot = launder_uniform(wslot);
        const int n = tile * WG_SAMPLES + wave * 32 + j;
        const bool valid = n < a.n_pts;
        const int nn = valid ? n : a.n_pts - 1;
        const float p[3] = {a.pts[3 * nn], a.pts[3 * nn + 1], a.pts[3 * nn + 2]};

        h8 xh[4], xl[4];      // X space fragments (lin0, colour lin0)
        float x3f[8];         // fp32 values of X k-step 3 (its pad slot carries a4[192] into lin4)
        {
            float f[4][8];
            encode_x(p, h, f);
#pragma unroll
            for (int s = 0; s < 4; ++s) split8(f[s], xh[s], xl[s]);
#pragma unroll
            for (int k = 0; k < 8; ++k) x3f[k] = f[3][k];
        }
        h8 ah[16], al[16], bh[16], bl[16];   // ping-pong activation fragments

        // ---- lin0: X -> a1 (2 chunks of 4 tiles) ------------------------------------------------
        static_for<2>([&](auto C) {
            constexpr int c = decltype(C)::value;
            const char* buf = ws.acquire(c == 0 ? CB_L0 : CB_HID);
            static_for<4>([&](auto TI) {
                constexpr int ti = decltype(TI)::value;
                constexpr int t = 4 * c + ti;
                f32x16 c1 = tail_tile(buf + 16 * KS_BYTES, ti, h), c2 = zero16();
                mma_tile<4, 0>(buf + ti * 4 * KS_BYTES, xh, xl, c1, c2, lane);
                f32x16 z = combine(c1, c2);
#pragma unroll
                for (int i = 0; i < 16; ++i) z[i] = softplus100(z[i]);
                if (FULL && !(a.dbg & 1)) stash_tile(slot(OS_A1 + 0), t, z, lane);
                split_tile(z, ah[2 * t], al[2 * t], ah[2 * t + 1], al[2 * t + 1]);
            });
        });

        struct Act {
            f32x16 v;
        };
        auto no_pre = [](auto, const char*) { return NoData{}; };
        // softplus layer: a_{l+1} = softplus(z) -> stash (reverse sweep) + next layer's fragments
        auto softplus_to = [&](h8(&oh)[16], h8(&ol)[16], int stash_slot) {
            return [&oh, &ol, stash_slot, &slot, lane, &a](auto T, f32x16 z, NoData) {
                constexpr int t = decltype(T)::value;
#pragma unroll
                for (int i = 0; i < 16; ++i) z[i] = softplus100(z[i]);
                if (FULL && !(a.dbg & 1)) stash_tile(slot(stash_slot), t, z, lane);
                split_tile(z, oh[2 * t], ol[2 * t], oh[2 * t + 1], ol[2 * t + 1]);
            };
        };

        run_layer<8, 16, true>(ws, CB_HID, CB_HID, ah, al, lane, h, no_pre, softplus_to(bh, bl, OS_A1 + 1));   // lin1
        run_layer<8, 16, true>(ws, CB_HID, CB_HID, bh, bl, lane, h, no_pre, softplus_to(ah, al, OS_A1 + 2));   // lin2
        // ---- lin3: 193 outputs = 7 tiles (tile 6 holds neuron 192 in row 0)
        float a4_192 = 0.f;
        run_layer<7, 16, true>(ws, CB_HID, CB_HID, ah, al, lane, h, no_pre, [&](auto T, f32x16 z, NoData) {
            constexpr int t = decltype(T)::value;
#pragma unroll
            for (int i = 0; i < 16; ++i) z[i] = softplus100(z[i]);
            if constexpr (t == 6) {
                // rows 193..223 are padding (zero weights, zero bias give softplus(0)): drop them
#pragma unroll
                for (int i = 0; i < 16; ++i) z[i] = (i == 0 && h == 0) ? z[i] : 0.f;
                a4_192 = z[0];
            }
            if (FULL && !(a.dbg & 1)) stash_tile(slot(OS_A1 + 3), t, z, lane);
            if constexpr (t < 6) split_tile(z, bh[2 * t], bl[2 * t], bh[2 * t + 1], bl[2 * t + 1]);
        });
        // ---- lin4: [a4 (192 via k-steps 0..11) | X with a4[192] in its pad slot] / sqrt2
        {
            const float v192 = __shfl_xor(a4_192, 32, 64);   // half 1 receives half 0's value
            float f3[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) f3[k] = x3f[k];
            f3[7] = h ? v192 : x3f[7];
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                bh[12 + s] = xh[s];
                bl[12 + s] = xl[s];
            }
            split8(f3, bh[15], bl[15]);
        }
        run_layer<8, 16, true>(ws, CB_HID, CB_HID, bh, bl, lane, h, no_pre, softplus_to(ah, al, OS_A1 + 4));   // lin4
        run_layer<8, 16, true>(ws, CB_HID, CB_HID, ah, al, lane, h, no_pre, softplus_to(bh, bl, OS_A1 + 5));   // lin5
        run_layer<8, 16, true>(ws, CB_HID, CB_HID, bh, bl, lane, h, no_pre, softplus_to(ah, al, OS_A1 + 6));   // lin6
        // ---- lin7 -> a8; sdf = W8[0,:] a8 + b8; seed of the reverse sweep dz7 = sigma'(z7) W8[0,:] / scale
        float sdf_acc = 0.f;
        run_layer<8, 16, true>(
            ws, CB_HID, FULL ? CB_HID : (more ? CB_L0 : 0), ah, al, lane, h,
            [&](auto, const char* tail) { return Act{tail_tile(tail, 1, h)}; },
            [&](auto T, f32x16 z, const Act& w8) {
                constexpr int t = decltype(T)::value;
                f32x16 dz;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    z[i] = softplus100(z[i]);
                    sdf_acc = fmaf(w8.v[i], z[i], sdf_acc);
                    dz[i] = dsoftplus_from_act(z[i]) * w8.v[i] * a.inv_scale;
                }
                if (FULL) {
                    split_tile(z, bh[2 * t], bl[2 * t], bh[2 * t + 1], bl[2 * t + 1]);   // a8 feeds lin8
                    h8 dh0, dl0, dh1, dl1;
                    split_tile(dz, dh0, dl0, dh1, dl1);
                    stash_frag(slot(OS_DZ7), 2 * t, dh0, dl0, lane);
                    stash_frag(slot(OS_DZ7), 2 * t + 1, dh1, dl1, lane);
                }
            });
        const float sdf = (half_sum(sdf_acc) + a.b8) * a.inv_scale;
        if (!FULL) {
            if (valid && h == 0) a.sdf[n] = sdf;
            continue;
        }

        // ---- lin8 rows 1..256: the feature vector (no activation) -> stash as fragments for colour lin0
        run_layer<8, 16, true>(ws, CB_HID, CB_BWD, bh, bl, lane, h, no_pre, [&](auto T, f32x16 z, NoData) {
            constexpr int t = decltype(T)::value;
            if (a.feat != nullptr && valid) {
#pragma unroll
                for (int i = 0; i < 16; ++i) a.feat[(size_t)n * H + 32 * t + tile_row(i, h)] = z[i];
            }
            h8 fh0, fl0, fh1, fl1;
            split_tile(z, fh0, fl0, fh1, fl1);
            stash_frag(slot(OS_FVEC), 2 * t, fh0, fl0, lane);
            stash_frag(slot(OS_FVEC), 2 * t + 1, fh1, fl1, lane);
        });

        // ---- reverse sweep: dz_{l-1} = sigma'(z_{l-1}) * (W_l^T dz_l); sigma' from the stashed activation a_l
        auto act_of = [&](int act_slot) {
            return [&slot, act_slot, lane, &a](auto T, const char*) {
                if (a.dbg & 2) return Act{zero16()};
                return Act{unstash_tile(slot(act_slot), decltype(T)::value, lane)};
            };
        };
        auto dsig_to = [&](h8(&oh)[16], h8(&ol)[16]) {
            return [&oh, &ol](auto T, f32x16 g, const Act& act) {
                constexpr int t = decltype(T)::value;
#pragma unroll
                for (int i = 0; i < 16; ++i) g[i] *= dsoftplus_from_act(act.v[i]);
                split_tile(g, oh[2 * t], ol[2 * t], oh[2 * t + 1], ol[2 * t + 1]);
            };
        };
#pragma unroll
        for (int s = 0; s < 16; ++s) unstash_frag(slot(OS_DZ7), s, ah[s], al[s], lane);
        run_layer<8, 16, false>(ws, CB_BWD, CB_BWD, ah, al, lane, h, act_of(OS_A1 + 6), dsig_to(bh, bl));   // W7^T -> dz6
        run_layer<8, 16, false>(ws, CB_BWD, CB_BWD, bh, bl, lane, h, act_of(OS_A1 + 5), dsig_to(ah, al));   // W6^T -> dz5
        run_layer<8, 16, false>(ws, CB_BWD, CB_BWD, ah, al, lane, h, act_of(OS_A1 + 4),                     // W5^T -> dz4 (kept)
                                [&](auto T, f32x16 g, const Act& act) {
                                    constexpr int t = decltype(T)::value;
#pragma unroll
                                    for (int i = 0; i < 16; ++i) g[i] *= dsoftplus_from_act(act.v[i]);
                                    split_tile(g, bh[2 * t], bl[2 * t], bh[2 * t + 1], bl[2 * t + 1]);
                                    stash_frag(slot(OS_DZ4), 2 * t, bh[2 * t], bl[2 * t], lane);
                                    stash_frag(slot(OS_DZ4), 2 * t + 1, bh[2 * t + 1], bl[2 * t + 1], lane);
                                });
        // W4[:, :193]^T: dz4 -> dz3 (193 rows = 7 tiles; a4's padding rows were stashed as 0 => sigma' = 0)
        run_layer<7, 16, false>(ws, CB_BWD, CB_BWD3, bh, bl, lane, h, act_of(OS_A1 + 3), [&](auto T, f32x16 g, const Act& act) {
            constexpr int t = decltype(T)::value;
#pragma unroll
            for (int i = 0; i < 16; ++i) g[i] *= dsoftplus_from_act(act.v[i]);
            if constexpr (t < 6) {
                split_tile(g, ah[2 * t], al[2 * t], ah[2 * t + 1], al[2 * t + 1]);
            } else {
                h8 d0, d1;   // only k-step 12 exists (neuron 192); 13 is padding
                split_tile(g, ah[12], al[12], d0, d1);
            }
        });
        run_layer<8, 13, false>(ws, CB_BWD3, CB_BWD, ah, al, lane, h, act_of(OS_A1 + 2), dsig_to(bh, bl));   // W3^T -> dz2
        run_layer<8, 16, false>(ws, CB_BWD, CB_BWD, bh, bl, lane, h, act_of(OS_A1 + 1), dsig_to(ah, al));    // W2^T -> dz1
        run_layer<8, 16, false>(ws, CB_BWD, CB_BWD, ah, al, lane, h, act_of(OS_A1 + 0), dsig_to(bh, bl));    // W1^T -> dz0
        // d sdf / d X-space = W0^T dz0 + W4[:, 193:]^T dz4   (64 rows = 2 tiles; row <-> k-slot of the same lane)
        f32x16 G1[2] = {zero16(), zero16()}, G2[2] = {zero16(), zero16()};
        static_for<2>([&](auto U) {
            constexpr int u = decltype(U)::value;
            const char* buf = ws.acquire(CB_BWD);
            mma_tile<16, 0>(buf, bh, bl, G1[u], G2[u], lane);
        });
#pragma unroll
        for (int s = 0; s < 16; ++s) unstash_frag(slot(OS_DZ4), s, ah[s], al[s], lane);
        static_for<2>([&](auto U) {
            constexpr int u = decltype(U)::value;
            const char* buf = ws.acquire(u == 0 ? CB_BWD : CB_C0A);
            mma_tile<16, 0>(buf, ah, al, G1[u], G2[u], lane);
        });
        // ---- Jacobian of the encoding (in-lane: G row of tile u, register 8(s&1)+j <-> k-slot (s = 2u + .., h, j))
        float g[3] = {0.f, 0.f, 0.f};
        {
            float f[4][8];
            encode_x(p, h, f);
            const f32x16 G0 = combine(G1[0], G2[0]), Gb = combine(G1[1], G2[1]);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float fr = 1.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float other = __shfl_xor(f[c][k], 32, 64);   // the conjugate function of the same angle
                    const float Gv = (c < 2) ? G0[8 * c + k] : Gb[k];
                    g[c] = fmaf(Gv, (h ? -fr : fr) * other, g[c]);
                    fr *= 2.f;
                }
            }
#pragma unroll
            for (int jj = 0; jj < 6; ++jj) {
                const float fr = (jj & 1) ? 512.f : 256.f;
                const float other = __shfl_xor(f[3][jj], 32, 64);
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
        const int ray = nn / a.spr;
        const float d[3] = {a.rays_d[3 * ray], a.rays_d[3 * ray + 1], a.rays_d[3 * ray + 2]};
        h8 mh[8], ml[8];   // the 8 k-steps of chunk B: X (4), enc(d) (2), enc(g) (2)
        {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                mh[s] = xh[s];
                ml[s] = xl[s];
            }
            float fd[2][8], fg[2][8];
            encode_v4(d, h, fd);
            encode_v4(g, h, fg);
            split8(fd[0], mh[4], ml[4]);
            split8(fd[1], mh[5], ml[5]);
            split8(fg[0], mh[6], ml[6]);
            split8(fg[1], mh[7], ml[7]);
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) unstash_frag(slot(OS_FVEC), s, ah[s], al[s], lane);
        static_for<8>([&](auto T) {
            constexpr int t = decltype(T)::value;
            const char* bufa = ws.acquire(CB_C0B);
            f32x16 c1 = zero16(), c2 = zero16();
            mma_tile<16, 0>(bufa, ah, al, c1, c2, lane);
            const char* bufb = ws.acquire(t + 1 < 8 ? CB_C0A : CB_HID);
            const f32x16 bias = tail_tile(bufb + 8 * KS_BYTES, 0, h);
            mma_tile<8, 0>(bufb, mh, ml, c1, c2, lane);
            f32x16 z = combine(c1, c2);
#pragma unroll
            for (int i = 0; i < 16; ++i) z[i] = fmaxf(z[i] + bias[i], 0.f);
            split_tile(z, bh[2 * t], bl[2 * t], bh[2 * t + 1], bl[2 * t + 1]);
        });
        auto relu_to = [&](h8(&oh)[16], h8(&ol)[16]) {
            return [&oh, &ol](auto T, f32x16 z, NoData) {
                constexpr int t = decltype(T)::value;
#pragma unroll
                for (int i = 0; i < 16; ++i) z[i] = fmaxf(z[i], 0.f);
                split_tile(z, oh[2 * t], ol[2 * t], oh[2 * t + 1], ol[2 * t + 1]);
            };
        };
        run_layer<8, 16, true>(ws, CB_HID, CB_HID, bh, bl, lane, h, no_pre, relu_to(ah, al));   // colour lin1
        run_layer<8, 16, true>(ws, CB_HID, CB_HID, ah, al, lane, h, no_pre, relu_to(bh, bl));   // colour lin2
        float rgb[3] = {0.f, 0.f, 0.f};
        struct W3 {
            f32x16 w[3];
        };
        run_layer<8, 16, true>(   // colour lin3 + the 3 rows of lin4 (tail slots 1..3)
            ws, CB_HID, more ? CB_L0 : 0, bh, bl, lane, h,
            [&](auto, const char* tail) { return W3{{tail_tile(tail, 1, h), tail_tile(tail, 2, h), tail_tile(tail, 3, h)}}; },
            [&](auto, f32x16 z, const W3& w) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int i = 0; i < 16; ++i) rgb[c] = fmaf(w.w[c][i], fmaxf(z[i], 0.f), rgb[c]);
            });
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[c] = sigmoid_fast(half_sum(rgb[c]) + a.c_blast[c]);
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

constexpr size_t OBJ2_LDS = 2 * CHUNK_MAX;

static int obj2_grid(int n_pts, int n_cus) {
    const int n_tiles = (n_pts + WG_SAMPLES - 1) / WG_SAMPLES;
    return n_tiles < n_cus ? n_tiles : n_cus;
}

size_t field2_obj_workspace_bytes(int n_pts, int n_cus) {
    return (size_t)obj2_grid(n_pts, n_cus) * WG_WAVES * OBJ2_SLOTS * SLOT_F4 * sizeof(float4);
}

int launch_field2_obj(const hn_field* f, const float* pts, const float* rays_d, int n_pts, int spr, float* sdf,
                      float* grad, float* rgb, float* feat, void* workspace, size_t workspace_bytes, bool full,
                      hipStream_t stream) {
    if (n_pts <= 0) return HN_OK;
    Obj2Args a;
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
    a.b8 = f->sdf_b8;
    for (int c = 0; c < 3; ++c) a.c_blast[c] = f->col_blast[c];
    a.sdf = sdf;
    a.grad = grad;
    a.rgb = rgb;
    a.feat = feat;
    a.scratch = reinterpret_cast<float4*>(workspace);
    {
        const char* e = getenv("HN_DBG");
        a.dbg = e ? atoi(e) : 0;
    }
    int n_cus = hn_device_cus();
    if (n_cus <= 0) n_cus = 256;
    const int grid = obj2_grid(n_pts, n_cus);
    if (full) {
        const size_t need = (size_t)grid * WG_WAVES * OBJ2_SLOTS * SLOT_F4 * sizeof(float4);
        if (workspace == nullptr || workspace_bytes < need) {
            set_error("field workspace too small: %zu < %zu", workspace_bytes, need);
            return HN_ENOMEM;
        }
    }
    static bool attr_set = false;
    if (!attr_set) {
        HN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_field2_obj<true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)OBJ2_LDS));
        HN_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_field2_obj<false>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)OBJ2_LDS));
        attr_set = true;
    }
    if (full)
        hipLaunchKernelGGL(k_field2_obj<true>, dim3(grid), dim3(256), OBJ2_LDS, stream, a);
    else
        hipLaunchKernelGGL(k_field2_obj<false>, dim3(grid), dim3(256), OBJ2_LDS, stream, a);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

}  // namespace v2
}  // namespace hn
