// Register-resident MLP building blocks for gfx950 (wave64, v_mfma_f32_32x32x2_f32).
//
// Layout: one wave owns 32 samples.  Every activation matrix is held TRANSPOSED,
// [neurons x 32 samples], as accumulator tiles of the 32x32x2 f32 MFMA: lane l
// holds sample (column) l&31, and register r of lane half h = l>>5 holds neuron
// (row) (r&3) + 8*(r>>2) + 4*h of the tile.  A layer is Y^T = W X^T: the weight
// fragment is the A operand (rows = output neurons), the previous layer's
// accumulator registers are fed back as the B operand with no data movement at
// all (k-step (u,r) contracts rows tile_row(r,0) and tile_row(r,1) of input
// tile u; the packed weights are laid out to match).  Nothing goes through LDS.
//
// Register budget: a layer keeps its whole INPUT (8 tiles = 128 VGPRs) in registers and
// streams the OUTPUT one 32-neuron tile at a time (16 accumulators) through a runtime loop;
// finished tiles go to a per-wave scratch slot in global memory (L2/MALL resident), which is
// where the reverse sweep and the fitting backward need the activations anyway.  That keeps
// the kernels under 256 VGPRs (2 waves per SIMD hide the slot round trip) and small in code.
#pragma once
#include "hn_common.h"

namespace hn {

using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ f32x16 mfma2(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ---- activations ---------------------------------------------------------------------------
// nn.Softplus(beta=100, threshold=20): utils/fields.py:125, 310.
__device__ __forceinline__ float softplus100(float z) {
    const float t = 100.f * z;
    const float y = __expf(fminf(t, 20.f));
    // log1p(y): series below 1e-3 (where 1+y would round away y's low bits)
    const float small = y * (1.f - y * (0.5f - y * 0.33333334f));
    const float big = __logf(1.f + y);
    const float r = (y < 1e-3f ? small : big) * 0.01f;
    return t > 20.f ? z : r;
}
// d softplus / dz = sigmoid(100 z), recovered from the activation a = softplus(z):
// exp(100 a) = 1 + exp(100 z)  =>  sigmoid(100 z) = 1 - exp(-100 a).
__device__ __forceinline__ float dsoftplus_from_act(float a) { return 1.f - __expf(-100.f * a); }
// d2 softplus / dz2 = 100 s (1 - s)
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// ---- accurate sin/cos of 2^k * x -----------------------------------------------------------
// The encodings evaluate sin/cos(2^k x), k < 10, on |x| up to a few units, i.e.
// arguments up to ~1e3 rad: sincosf's full-range reduction is the accurate and
// simple choice (the fast __sinf loses absolute accuracy with |arg|).
__device__ __forceinline__ void sincos_acc(float x, float* s, float* c) { sincosf(x, s, c); }

// ---- tile helpers --------------------------------------------------------------------------
// bias in tile-row order: lane half h, registers 4q..4q+3 <- b[32 t + 8 q + 4 h + (0..3)]
__device__ __forceinline__ f32x16 load_bias_tile(const float* __restrict__ b, int t, int h) {
    f32x16 acc;
    const float4* p = reinterpret_cast<const float4*>(b + 32 * t + 4 * h);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = p[2 * q];
        acc[4 * q + 0] = v.x;
        acc[4 * q + 1] = v.y;
        acc[4 * q + 2] = v.z;
        acc[4 * q + 3] = v.w;
    }
    return acc;
}

__device__ __forceinline__ f32x16 zero_tile() {
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    return acc;
}

// acc += W[tile rows, 32 k-columns of input tile] * x   (16 k-steps, 4 fragment loads)
// w points at the float4 fragment [step/4 = 0][lane 0] of this (out tile, input tile) block.
__device__ __forceinline__ void mma_tile(f32x16& acc, const float4* __restrict__ w, const f32x16& x, int lane) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 a = w[q * 64 + lane];
        acc = mfma2(a.x, x[4 * q + 0], acc);
        acc = mfma2(a.y, x[4 * q + 1], acc);
        acc = mfma2(a.z, x[4 * q + 2], acc);
        acc = mfma2(a.w, x[4 * q + 3], acc);
    }
}

// acc += W[tile rows, 4*NQ k-steps] * b[0..4*NQ)   where b[] are per-lane B values
template <int NQ>
__device__ __forceinline__ void mma_steps(f32x16& acc, const float4* __restrict__ w, const float (&b)[4 * NQ],
                                          int lane) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const float4 a = w[q * 64 + lane];
        acc = mfma2(a.x, b[4 * q + 0], acc);
        acc = mfma2(a.y, b[4 * q + 1], acc);
        acc = mfma2(a.z, b[4 * q + 2], acc);
        acc = mfma2(a.w, b[4 * q + 3], acc);
    }
}

// Y[t] (+)= W x for a hidden layer with KT input tiles and OT output tiles.
// Packed weights: [OT][KT*4][64] float4.
// Arrays may be larger than OT / KT (only the first OT / KT tiles are touched).
template <int OT, int KT, int NY, int NX>
__device__ __forceinline__ void dense_from_tiles(f32x16 (&y)[NY], const float4* __restrict__ w, const f32x16 (&x)[NX],
                                                 int lane) {
    static_assert(OT <= NY && KT <= NX, "tile counts");
#pragma unroll
    for (int t = 0; t < OT; ++t) {
#pragma unroll
        for (int u = 0; u < KT; ++u) mma_tile(y[t], w + (size_t)(t * KT + u) * 4 * 64, x[u], lane);
    }
}

template <int OT, int NY>
__device__ __forceinline__ void init_bias(f32x16 (&y)[NY], const float* __restrict__ b, int h) {
#pragma unroll
    for (int t = 0; t < OT; ++t) y[t] = load_bias_tile(b, t, h);
}

template <int OT, int NY>
__device__ __forceinline__ void apply_softplus(f32x16 (&y)[NY]) {
#pragma unroll
    for (int t = 0; t < OT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) y[t][i] = softplus100(y[t][i]);
}

template <int OT, int NY>
__device__ __forceinline__ void apply_relu(f32x16 (&y)[NY]) {
#pragma unroll
    for (int t = 0; t < OT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) y[t][i] = fmaxf(y[t][i], 0.f);
}

// ---- per-wave scratch (global memory, L2/MALL resident): slots of OT tiles ------------------
// slot layout: [tile][q][lane] float4 -> every store/load instruction moves 1 KiB contiguous.
template <int OT, int NY>
__device__ __forceinline__ void store_tiles(float4* __restrict__ slot, const f32x16 (&y)[NY], int lane) {
#pragma unroll
    for (int t = 0; t < OT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            slot[(t * 4 + q) * 64 + lane] = make_float4(y[t][4 * q], y[t][4 * q + 1], y[t][4 * q + 2], y[t][4 * q + 3]);
}
__device__ __forceinline__ f32x16 load_tile(const float4* __restrict__ slot, int t, int lane) {
    f32x16 y;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = slot[(t * 4 + q) * 64 + lane];
        y[4 * q] = v.x;
        y[4 * q + 1] = v.y;
        y[4 * q + 2] = v.z;
        y[4 * q + 3] = v.w;
    }
    return y;
}
constexpr size_t SLOT_FLOAT4 = (size_t)NT * 4 * 64;   // float4 per 8-tile slot (32 KiB)

__device__ __forceinline__ void store_tile(float4* __restrict__ slot, int t, const f32x16& y, int lane) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
        slot[(t * 4 + q) * 64 + lane] = make_float4(y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
}
template <int KT>
__device__ __forceinline__ void load_tiles(f32x16 (&x)[KT], const float4* __restrict__ slot, int lane) {
#pragma unroll
    for (int u = 0; u < KT; ++u) x[u] = load_tile(slot, u, lane);
}

enum { ACT_NONE = 0, ACT_SOFTPLUS = 1, ACT_RELU = 2 };
template <int ACT>
__device__ __forceinline__ void activate(f32x16& y) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (ACT == ACT_SOFTPLUS) y[i] = softplus100(y[i]);
        if (ACT == ACT_RELU) y[i] = fmaxf(y[i], 0.f);
    }
}

struct NoExtra {
    __device__ __forceinline__ void operator()(f32x16&, int) const {}
};

// out_slot[t] = act(bias[t] + W[t, :] x + extra(t)),  x = the KT tiles already in registers.
// Runtime loop over the OT output tiles: 16 live accumulators.
template <int OT, int KT, int ACT, typename Extra>
__device__ __forceinline__ void layer_from_regs(const float4* __restrict__ w, const float* __restrict__ bias,
                                                const f32x16 (&x)[KT], float4* __restrict__ out_slot, int lane, int h,
                                                Extra extra) {
#pragma unroll 1
    for (int t = 0; t < OT; ++t) {
        f32x16 acc = bias != nullptr ? load_bias_tile(bias, t, h) : zero_tile();
        const float4* wt = w + (size_t)t * KT * 4 * 64;
#pragma unroll
        for (int u = 0; u < KT; ++u) mma_tile(acc, wt + (size_t)u * 4 * 64, x[u], lane);
        extra(acc, t);
        activate<ACT>(acc);
        store_tile(out_slot, t, acc, lane);
    }
}

// the same with the input read from a slot first
template <int OT, int KT, int ACT, typename Extra>
__device__ __forceinline__ void layer_slots(const float4* __restrict__ w, const float* __restrict__ bias,
                                            const float4* __restrict__ in_slot, float4* __restrict__ out_slot, int lane,
                                            int h, Extra extra) {
    f32x16 x[KT];
    load_tiles<KT>(x, in_slot, lane);
    layer_from_regs<OT, KT, ACT>(w, bias, x, out_slot, lane, h, extra);
}

// reverse-sweep step: dz_prev[t] = sigma'(z_prev)[t] * (W^T dz)[t], sigma' recovered from the saved activation
template <int OT, int KT>
__device__ __forceinline__ void layer_bwd_slots(const float4* __restrict__ wT, const float4* __restrict__ dz_slot,
                                                const float4* __restrict__ act_slot, float4* __restrict__ out_slot,
                                                int lane) {
    f32x16 x[KT];
    load_tiles<KT>(x, dz_slot, lane);
#pragma unroll 1
    for (int t = 0; t < OT; ++t) {
        f32x16 acc = zero_tile();
        const float4* wt = wT + (size_t)t * KT * 4 * 64;
#pragma unroll
        for (int u = 0; u < KT; ++u) mma_tile(acc, wt + (size_t)u * 4 * 64, x[u], lane);
        const f32x16 act = load_tile(act_slot, t, lane);
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] *= dsoftplus_from_act(act[i]);
        store_tile(out_slot, t, acc, lane);
    }
}

// sum over the two lane halves (the two halves hold complementary rows of a column)
__device__ __forceinline__ float half_sum(float v) { return v + __shfl_xor(v, 32, 64); }

// dot(w[neuron], x[neuron]) over a full 8-tile activation for this lane's sample:
// w is a plain [256] vector; each lane touches its own 128 rows, then halves are added.
template <int KT, int NX>
__device__ __forceinline__ float row_dot(const float* __restrict__ w, const f32x16 (&x)[NX], int h) {
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < KT; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(w + 32 * u + 8 * q + 4 * h);
            s = fmaf(v.x, x[u][4 * q + 0], s);
            s = fmaf(v.y, x[u][4 * q + 1], s);
            s = fmaf(v.z, x[u][4 * q + 2], s);
            s = fmaf(v.w, x[u][4 * q + 3], s);
        }
    return half_sum(s);
}

}  // namespace hn
