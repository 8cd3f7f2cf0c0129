// Object field: SDFNetwork_OBJ + analytic d sdf / d p + RenderingNetwork_OBJ, fused, one
// wave per 32 samples, activations never leave the register file (see hn_mlp.h).
//
// Reference: utils/fields.py:316-347 (sdf net, .gradient), :387-405 (colour net),
// called from utils/renderer.py:130-135 / 380-385.
#include "hn_mlp.h"

namespace hn {

struct FieldObjArgs {
    const float* pts;       // [n,3]
    const float* rays_d;    // [n/spr,3]
    int n_pts;
    int spr;                // samples per ray
    float inv_scale;        // 1/scale applied to the sdf output (utils/fields.py:328)
    // packed network
    const float4* w_fwd[9];   // l=0: over X space (32 steps); 1..7 hidden; 8: feature rows
    const float4* w_skip;     // W4 skip columns over X space
    const float* bias[9];
    const float* w8row;
    float b8;
    const float4* w_bwd[8];   // l=1..7
    const float4* w_bwd_in0;  // [2 tiles][8*4]
    const float4* w_bwd_in4;  // [2 tiles][8*4]
    const float4* c_in_x;     // [8][8]   (32 steps)
    const float4* c_in_d;     // [8][4]   (16 steps)
    const float4* c_in_f;     // [8][32]
    const float4* c_in_g;     // [8][4]
    const float4* c_fwd[4];   // 1..3
    const float* c_bias[4];
    const float* c_wlast;     // [3][256]
    float c_blast[3];
    // outputs
    float* sdf;
    float* grad;
    float* rgb;
    float* feat;    // optional [n,256]
    float4* scratch;  // per-wave slots, see the OS_* enum
};

// B values of the X space for this lane: step s < 30: (sin, cos)(2^(s%10) p_(s/10));
// 30: (px, py); 31: (pz, 0).  Lane half 0 takes the first member, half 1 the second.
__device__ __forceinline__ void encode_obj_x(const float p[3], int h, float (&bx)[OBJ_X_STEPS]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float f = 1.f;
#pragma unroll
        for (int k = 0; k < PTS_FREQS; ++k) {
            float s, co;
            sincos_acc(p[c] * f, &s, &co);
            bx[c * PTS_FREQS + k] = h ? co : s;
            f *= 2.f;
        }
    }
    bx[30] = h ? p[1] : p[0];
    bx[31] = h ? 0.f : p[2];
}

// enc4 of a 3-vector in VEC space: steps 0..11: (sin, cos)(2^(s%4) v_(s/4)); 12: (x,y); 13: (z,0); 14,15 pad
__device__ __forceinline__ void encode_vec4(const float v[3], int h, float (&b)[VEC_STEPS]) {
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

// scratch slots of one wave (32 KiB each)
enum {
    OS_A1 = 0,       // a1..a8 -> slots 0..7 (activation of layer l in slot l-1)
    OS_C0 = 8,       // colour lin0 pre-activation (without / then with the gradient columns)
    OS_DZ = 9,       // dz ping-pong: 9, 10
    OS_DZ4 = 11,     // dz4 is needed twice
    OS_COL = 9,      // colour hidden activations reuse the dz slots
    OBJ_SLOTS_FULL = 12,
    OBJ_SLOTS_SDF = 2,
};

template <bool FULL>
__global__ __launch_bounds__(64, 2) void k_field_obj(const FieldObjArgs a) {
    const int lane = threadIdx.x;
    const int j = lane & 31;
    const int h = lane >> 5;
    float4* const base = a.scratch + (size_t)blockIdx.x * (FULL ? OBJ_SLOTS_FULL : OBJ_SLOTS_SDF) * SLOT_FLOAT4;
    auto slot = [&](int i) { return base + (size_t)i * SLOT_FLOAT4; };
    // activation of layer l: FULL keeps them all, sdf-only ping-pongs
    auto act_slot = [&](int l) { return slot(FULL ? l - 1 : (l & 1)); };
    const int n_tiles = (a.n_pts + 31) / 32;

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int n = tile * 32 + j;
        const bool valid = n < a.n_pts;
        const int nn = valid ? n : a.n_pts - 1;
        float p[3] = {a.pts[3 * nn], a.pts[3 * nn + 1], a.pts[3 * nn + 2]};
        float bx[OBJ_X_STEPS];
        encode_obj_x(p, h, bx);

        // ---- lin0: X space -> a1
#pragma unroll 1
        for (int t = 0; t < NT; ++t) {
            f32x16 acc = load_bias_tile(a.bias[0], t, h);
            mma_steps<8>(acc, a.w_fwd[0] + (size_t)t * 8 * 64, bx, lane);
            activate<ACT_SOFTPLUS>(acc);
            store_tile(act_slot(1), t, acc, lane);
        }
        // ---- lin1, lin2
        layer_slots<NT, NT, ACT_SOFTPLUS>(a.w_fwd[1], a.bias[1], act_slot(1), act_slot(2), lane, h, NoExtra());
        layer_slots<NT, NT, ACT_SOFTPLUS>(a.w_fwd[2], a.bias[2], act_slot(2), act_slot(3), lane, h, NoExtra());
        // ---- lin3: 193 outputs = 7 tiles
        layer_slots<7, NT, ACT_SOFTPLUS>(a.w_fwd[3], a.bias[3], act_slot(3), act_slot(4), lane, h, NoExtra());
        // ---- lin4: [a4 (7 tiles), X]/sqrt2   (1/sqrt2 folded into the packed weights)
        layer_slots<NT, 7, ACT_SOFTPLUS>(a.w_fwd[4], a.bias[4], act_slot(4), act_slot(5), lane, h,
                                         [&](f32x16& acc, int t) {
                                             mma_steps<8>(acc, a.w_skip + (size_t)t * 8 * 64, bx, lane);
                                         });
        // ---- lin5..lin7
        layer_slots<NT, NT, ACT_SOFTPLUS>(a.w_fwd[5], a.bias[5], act_slot(5), act_slot(6), lane, h, NoExtra());
        layer_slots<NT, NT, ACT_SOFTPLUS>(a.w_fwd[6], a.bias[6], act_slot(6), act_slot(7), lane, h, NoExtra());
        layer_slots<NT, NT, ACT_SOFTPLUS>(a.w_fwd[7], a.bias[7], act_slot(7), act_slot(8), lane, h, NoExtra());
        // ---- lin8: row 0 = sdf; rows 1..256 = feature vector, fed straight into colour lin0
        float sdf;
        {
            f32x16 x[NT];
            load_tiles<NT>(x, act_slot(8), lane);
            sdf = (row_dot<NT>(a.w8row, x, h) + a.b8) * a.inv_scale;
            if (!FULL) {
                if (valid && h == 0) a.sdf[n] = sdf;
                continue;
            }
            layer_from_regs<NT, NT, ACT_NONE>(a.w_fwd[8], a.bias[8], x, slot(OS_DZ), lane, h, NoExtra());
        }
        if (a.feat != nullptr && valid) {
#pragma unroll 1
            for (int t = 0; t < NT; ++t) {
                const f32x16 f = load_tile(slot(OS_DZ), t, lane);
#pragma unroll
                for (int r = 0; r < 16; ++r) a.feat[(size_t)n * H + 32 * t + tile_row(r, h)] = f[r];
            }
        }
        // ---- colour lin0 without the gradient columns: [enc(p) | enc(d) | feature] -> slot C0
        const int ray = nn / a.spr;
        float d[3] = {a.rays_d[3 * ray], a.rays_d[3 * ray + 1], a.rays_d[3 * ray + 2]};
        {
            float bd[VEC_STEPS];
            encode_vec4(d, h, bd);
            layer_slots<NT, NT, ACT_NONE>(a.c_in_f, a.c_bias[0], slot(OS_DZ), slot(OS_C0), lane, h,
                                          [&](f32x16& acc, int t) {
                                              mma_steps<8>(acc, a.c_in_x + (size_t)t * 8 * 64, bx, lane);
                                              mma_steps<4>(acc, a.c_in_d + (size_t)t * 4 * 64, bd, lane);
                                          });
        }
        // ---- reverse sweep: dz7 = sigma'(z7) * W8[0,:] / scale
#pragma unroll 1
        for (int t = 0; t < NT; ++t) {
            const f32x16 act = load_tile(act_slot(8), t, lane);
            f32x16 dz;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 w = *reinterpret_cast<const float4*>(a.w8row + 32 * t + 8 * q + 4 * h);
                dz[4 * q + 0] = dsoftplus_from_act(act[4 * q + 0]) * w.x * a.inv_scale;
                dz[4 * q + 1] = dsoftplus_from_act(act[4 * q + 1]) * w.y * a.inv_scale;
                dz[4 * q + 2] = dsoftplus_from_act(act[4 * q + 2]) * w.z * a.inv_scale;
                dz[4 * q + 3] = dsoftplus_from_act(act[4 * q + 3]) * w.w * a.inv_scale;
            }
            store_tile(slot(OS_DZ + 1), t, dz, lane);   // dz7 -> slot 10 (odd)
        }
        // dz_l lives in slot OS_DZ + (l & 1), except dz4 in OS_DZ4
        layer_bwd_slots<NT, NT>(a.w_bwd[7], slot(OS_DZ + 1), act_slot(7), slot(OS_DZ + 0), lane);   // dz6
        layer_bwd_slots<NT, NT>(a.w_bwd[6], slot(OS_DZ + 0), act_slot(6), slot(OS_DZ + 1), lane);   // dz5
        layer_bwd_slots<NT, NT>(a.w_bwd[5], slot(OS_DZ + 1), act_slot(5), slot(OS_DZ4), lane);      // dz4
        layer_bwd_slots<7, NT>(a.w_bwd[4], slot(OS_DZ4), act_slot(4), slot(OS_DZ + 1), lane);       // dz3 (7 tiles)
        layer_bwd_slots<NT, 7>(a.w_bwd[3], slot(OS_DZ + 1), act_slot(3), slot(OS_DZ + 0), lane);    // dz2
        layer_bwd_slots<NT, NT>(a.w_bwd[2], slot(OS_DZ + 0), act_slot(2), slot(OS_DZ + 1), lane);   // dz1
        layer_bwd_slots<NT, NT>(a.w_bwd[1], slot(OS_DZ + 1), act_slot(1), slot(OS_DZ + 0), lane);   // dz0
        // d sdf / d X-space inputs = W0^T dz0 + W4x^T dz4 (64 rows = 2 tiles)
        f32x16 G[2];
        G[0] = zero_tile();
        G[1] = zero_tile();
        {
            f32x16 x[NT];
            load_tiles<NT>(x, slot(OS_DZ + 0), lane);
            dense_from_tiles<2, NT>(G, a.w_bwd_in0, x, lane);
            load_tiles<NT>(x, slot(OS_DZ4), lane);
            dense_from_tiles<2, NT>(G, a.w_bwd_in4, x, lane);
        }
        // ---- Jacobian of the encoding: G rows (pair s = 16u + r, member h) -> d sdf / d p
        float g[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float f = 1.f;
#pragma unroll
            for (int k = 0; k < PTS_FREQS; ++k) {
                const int s = c * PTS_FREQS + k;
                // this lane holds sin (h=0) or cos (h=1); the derivative needs the other one
                const float other = __shfl_xor(bx[s], 32, 64);
                const float dval = h ? -f * other : f * other;
                g[c] = fmaf(G[s >> 4][s & 15], dval, g[c]);
                f *= 2.f;
            }
        }
        g[0] += h ? 0.f : G[1][14];
        g[1] += h ? G[1][14] : 0.f;
        g[2] += h ? 0.f : G[1][15];
        g[0] = half_sum(g[0]);
        g[1] = half_sum(g[1]);
        g[2] = half_sum(g[2]);
        // ---- colour: add the gradient columns, relu, lin1..lin3, lin4 + sigmoid
        {
            float bg[VEC_STEPS];
            encode_vec4(g, h, bg);
#pragma unroll 1
            for (int t = 0; t < NT; ++t) {
                f32x16 acc = load_tile(slot(OS_C0), t, lane);
                mma_steps<4>(acc, a.c_in_g + (size_t)t * 4 * 64, bg, lane);
                activate<ACT_RELU>(acc);
                store_tile(slot(OS_COL), t, acc, lane);
            }
        }
        layer_slots<NT, NT, ACT_RELU>(a.c_fwd[1], a.c_bias[1], slot(OS_COL), slot(OS_COL + 1), lane, h, NoExtra());
        layer_slots<NT, NT, ACT_RELU>(a.c_fwd[2], a.c_bias[2], slot(OS_COL + 1), slot(OS_COL), lane, h, NoExtra());
        layer_slots<NT, NT, ACT_RELU>(a.c_fwd[3], a.c_bias[3], slot(OS_COL), slot(OS_COL + 1), lane, h, NoExtra());
        float rgb[3];
        {
            f32x16 x[NT];
            load_tiles<NT>(x, slot(OS_COL + 1), lane);
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

int launch_field_obj(const hn_field* f, const float* pts, const float* rays_d, int n_pts, int spr, float* sdf,
                     float* grad, float* rgb, float* feat, void* workspace, size_t workspace_bytes, bool full,
                     hipStream_t stream);

size_t field_obj_workspace_bytes(int n_pts, int n_cus) {
    return (size_t)field_grid(n_pts, n_cus) * OBJ_SLOTS_FULL * SLOT_FLOAT4 * sizeof(float4);
}

int launch_field_obj(const hn_field* f, const float* pts, const float* rays_d, int n_pts, int spr, float* sdf,
                     float* grad, float* rgb, float* feat, void* workspace, size_t workspace_bytes, bool full,
                     hipStream_t stream) {
    if (n_pts <= 0) return HN_OK;
    FieldObjArgs a;
    a.pts = pts;
    a.rays_d = rays_d;
    a.n_pts = n_pts;
    a.spr = spr > 0 ? spr : 1;
    a.inv_scale = 1.f / f->scale;
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
    a.c_in_d = f->col_in_d.w;
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
    const size_t need = (size_t)grid * (full ? OBJ_SLOTS_FULL : OBJ_SLOTS_SDF) * SLOT_FLOAT4 * sizeof(float4);
    if (workspace == nullptr || workspace_bytes < need) {
        set_error("field workspace too small: %zu < %zu", workspace_bytes, need);
        return HN_ENOMEM;
    }
    if (full)
        hipLaunchKernelGGL(k_field_obj<true>, dim3(grid), dim3(64), 0, stream, a);
    else
        hipLaunchKernelGGL(k_field_obj<false>, dim3(grid), dim3(64), 0, stream, a);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

}  // namespace hn
