// v2 MLP building blocks for gfx950: fp16 hi/lo split operands on v_mfma_f32_32x32x16_f16
// ("f16x3": fp32-equivalent products at 16/3 x the f32-MFMA rate), weights streamed
// global -> LDS by DMA (global_load_lds_dwordx4) and shared by the 4 waves of a workgroup,
// activations resident in registers as MFMA B operands.
//
// Numerics.  x = hi + lo/2048 with hi = fp16(x) and lo = fp16((x - hi) * 2048): 22 mantissa
// bits, no range problem for the small second term.  A product W x is evaluated as
//   C1 += Whi xhi            C2 += Whi xlo + Wlo xhi            W x ~= C1 + C2 / 2048
// in fp32 accumulators (the lo*lo term, 2^-22 relative, is dropped): measured 5.6e-7 of the
// result's max on the reference's layers, the same as an fp32 matmul (5.5e-7).
//
// Layout.  One wave owns 32 samples; a workgroup is 4 waves = 128 samples and one workgroup
// runs per CU (512-register kernel).  Activations are [neurons x samples] accumulator tiles of
// the 32x32 MFMA: lane l holds column (sample) l&31; register i of lane half h = l>>5 holds row
// (i&3) + 8 (i>>2) + 4 h.  Registers 8u..8u+7 of tile t, converted to fp16, ARE the B fragment of
// k-step 2t+u of the next layer (element j <-> neuron 32t + 16u + 8 (j>>2) + 4h + (j&3)); the
// packed weights use the same k order, so nothing moves between lanes or through LDS.
//
// Weight stream.  The packer (hn_pack2.hip) lays every matrix out as CHUNKS in the exact order
// the kernel consumes them: chunk = TILES x KS k-step blocks of 2 KiB ([hi fragment 1 KiB][lo
// fragment 1 KiB], lane-linear, so one ds_read_b128 per fragment is conflict-free) + an optional
// 1 KiB tail of fp32 side data (bias of the tile, last-layer rows).  Two LDS buffers: the DMA of
// chunk i+1 is issued right after the barrier that publishes chunk i.
#pragma once
#include <type_traits>
#include <utility>

#include "hn_common.h"

namespace hn {
namespace v2 {

using h8 = _Float16 __attribute__((ext_vector_type(8)));
using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

constexpr int KS_BYTES = 2048;            // one k-step block: hi fragment + lo fragment
constexpr int TAIL_BYTES = 1024;          // 256 floats of side data
constexpr int CHUNK_MAX = 16 * KS_BYTES + TAIL_BYTES;   // 33 KiB
constexpr float LO_SCALE = 2048.f;
constexpr float LO_INV = 1.f / 2048.f;
constexpr int WG_WAVES = 4;
constexpr int WG_SAMPLES = 32 * WG_WAVES;

__host__ __device__ constexpr int chunk_bytes(int tiles, int ks, bool tail) {
    return tiles * ks * KS_BYTES + (tail ? TAIL_BYTES : 0);
}

// ---- weight stream ---------------------------------------------------------------------------
struct WStream {
    const char* g;      // global address of the next chunk to fetch
    const char* begin;  // the stream is cyclic: after the last chunk of a sample tile comes the first again
    const char* end;
    char* lds;          // two buffers of CHUNK_MAX bytes
    int phase;          // buffer that receives the next fetch
    int wave, lane;

    __device__ __forceinline__ void fetch(int bytes) {
        if (g == end) g = begin;
        char* dst = lds + phase * CHUNK_MAX;
        const int pieces = bytes >> 10;
        for (int p = wave; p < pieces; p += WG_WAVES)
            __builtin_amdgcn_global_load_lds((glb_void_t*)(g + (size_t)p * 1024 + lane * 16), (lds_void_t*)(dst + p * 1024),
                                             16, 0, 0);
        g += bytes;
        phase ^= 1;
    }
    // Publishes the chunk fetched last (all waves' pieces) and starts the fetch of the following
    // one (next_bytes == 0: nothing to fetch).  Returns the LDS address of the published chunk.
    __device__ __forceinline__ const char* acquire(int next_bytes) {
        __syncthreads();   // s_waitcnt vmcnt(0) + s_barrier: my pieces landed, everybody is done with the other buffer
        const char* cur = lds + (phase ^ 1) * CHUNK_MAX;
        if (next_bytes > 0) fetch(next_bytes);
        return cur;
    }
};

// ---- fragments -------------------------------------------------------------------------------
// fp16 hi / scaled-lo split of 8 fp32 values (one B fragment)
__device__ __forceinline__ void split8(const float (&x)[8], h8& hi, h8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const _Float16 hj = (_Float16)x[j];
        hi[j] = hj;
        lo[j] = (_Float16)((x[j] - (float)hj) * LO_SCALE);
    }
}
// registers 8u..8u+7 of an accumulator tile -> B fragment of k-step 2t+u
__device__ __forceinline__ void split_tile(const f32x16& a, h8& hi0, h8& lo0, h8& hi1, h8& lo1) {
    const float x0[8] = {a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7]};
    const float x1[8] = {a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15]};
    split8(x0, hi0, lo0);
    split8(x1, hi1, lo1);
}
// fp32 value back from a stored fragment element
__device__ __forceinline__ float unsplit(_Float16 hi, _Float16 lo) { return fmaf((float)lo, LO_INV, (float)hi); }

__device__ __forceinline__ f32x16 mfma16(const h8& a, const h8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// c1/c2 += W[tile rows, KS k-steps] * x[S0 .. S0+KS)  -- `blk` points at the tile's first k-step block in LDS
template <int KS, int S0, int NX>
__device__ __forceinline__ void mma_tile(const char* blk, const h8 (&xh)[NX], const h8 (&xl)[NX], f32x16& c1, f32x16& c2,
                                         int lane) {
    static_assert(S0 + KS <= NX, "k-step range");
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const h8 ah = *reinterpret_cast<const h8*>(blk + s * KS_BYTES + lane * 16);
        const h8 al = *reinterpret_cast<const h8*>(blk + s * KS_BYTES + 1024 + lane * 16);
        c1 = mfma16(ah, xh[S0 + s], c1);
        c2 = mfma16(ah, xl[S0 + s], c2);
        c2 = mfma16(al, xh[S0 + s], c2);
    }
}

// tail helpers: 32 floats stored [half][16] so that lane half h reads its 16 rows as 4 float4
// (row of register i, half h: (i&3) + 8 (i>>2) + 4 h)
__device__ __forceinline__ f32x16 tail_tile(const char* tail, int slot, int h) {
    f32x16 v;
    const float4* p = reinterpret_cast<const float4*>(tail + slot * 128 + h * 64);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 w = p[q];
        v[4 * q] = w.x;
        v[4 * q + 1] = w.y;
        v[4 * q + 2] = w.z;
        v[4 * q + 3] = w.w;
    }
    return v;
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 v;
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = 0.f;
    return v;
}
__device__ __forceinline__ f32x16 combine(const f32x16& c1, const f32x16& c2) {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = fmaf(c2[i], LO_INV, c1[i]);
    return z;
}

// ---- activations -----------------------------------------------------------------------------
// nn.Softplus(beta=100, threshold=20) (utils/fields.py:125, 310): max(z,0) + log1p(exp(-100|z|))/100.
// Beyond the threshold the log term is below half an ulp of z, i.e. the result IS z, as in torch.
__device__ __forceinline__ float softplus100(float z) {
    // branch-free, 6 VALU ops: exp(-100|z|) -> log2(1 + e) * ln2/100 + max(z, 0).  The rounding of
    // 1 + e costs at most 6e-8 * ln2/100 = 4e-10 absolute, far below the fp32 resolution of the sums
    // this value enters (and 4e-8 on the derivative recovered from it).
    const float e = __builtin_amdgcn_exp2f(-fabsf(z * 144.26950408889634f));
    return fmaf(__builtin_amdgcn_logf(1.f + e), 0.0069314718055994531f, fmaxf(z, 0.f));
}
// sigmoid(100 z) recovered from a = softplus(z): exp(100 a) = 1 + exp(100 z) => s = 1 - exp(-100 a)
__device__ __forceinline__ float dsoftplus_from_act(float a) {
    return 1.f - __builtin_amdgcn_exp2f(a * -144.26950408889634f);
}
__device__ __forceinline__ float sigmoid_fast(float x) { return 1.f / (1.f + __expf(-x)); }

// ---- a layer as a software pipeline over its output tiles ----------------------------------------
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}
struct NoData {};

// OT output tiles, one chunk of KS k-steps (+ 1 KiB tail if TAIL, whose slot 0 is the tile's bias) each.
// Step t: publish chunk t (barrier), start the DMA of the next one, fetch what tile t's epilogue will
// need (`pre`, e.g. side data from the tail, stashed activations), issue the 3 KS MFMAs of tile t, and --
// in the same scheduling region, so that VALU and MFMA overlap -- run the epilogue `post` of tile t-1.
template <int OT, int KS, bool TAIL, typename Pre, typename Post>
__device__ __forceinline__ void run_layer(WStream& ws, int cb_same, int next_after, const h8 (&xh)[16], const h8 (&xl)[16],
                                          int lane, int h, Pre&& pre, Post&& post) {
    using I0 = std::integral_constant<int, 0>;
    using PD = decltype(pre(I0{}, (const char*)nullptr));
    f32x16 c1[2], c2[2];
    PD pd[2];
    static_for<OT>([&](auto T) {
        constexpr int t = decltype(T)::value;
        const char* buf = ws.acquire(t + 1 < OT ? cb_same : next_after);
        const char* tail = buf + KS * KS_BYTES;
        c1[t & 1] = TAIL ? tail_tile(tail, 0, h) : zero16();
        c2[t & 1] = zero16();
        pd[t & 1] = pre(T, tail);
        mma_tile<KS, 0>(buf, xh, xl, c1[t & 1], c2[t & 1], lane);
        if constexpr (t > 0)
            post(std::integral_constant<int, t - 1>{}, combine(c1[(t - 1) & 1], c2[(t - 1) & 1]), pd[(t - 1) & 1]);
    });
    post(std::integral_constant<int, OT - 1>{}, combine(c1[(OT - 1) & 1], c2[(OT - 1) & 1]), pd[(OT - 1) & 1]);
}

// ---- per-wave stash in global memory (slots of 32 KiB, every instruction moves 1 KiB) -----------
constexpr size_t SLOT_F4 = 8 * 4 * 64;   // float4 per slot
__device__ __forceinline__ void stash_tile(float4* slot, int t, const f32x16& y, int lane) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
        slot[(t * 4 + q) * 64 + lane] = make_float4(y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
}
__device__ __forceinline__ f32x16 unstash_tile(const float4* slot, int t, int lane) {
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
// fragment stash: [k-step][hi|lo][lane] 16 B
__device__ __forceinline__ void stash_frag(float4* slot, int s, const h8& hi, const h8& lo, int lane) {
    slot[(s * 2) * 64 + lane] = *reinterpret_cast<const float4*>(&hi);
    slot[(s * 2 + 1) * 64 + lane] = *reinterpret_cast<const float4*>(&lo);
}
__device__ __forceinline__ void unstash_frag(const float4* slot, int s, h8& hi, h8& lo, int lane) {
    const float4 a = slot[(s * 2) * 64 + lane];
    const float4 b = slot[(s * 2 + 1) * 64 + lane];
    hi = *reinterpret_cast<const h8*>(&a);
    lo = *reinterpret_cast<const h8*>(&b);
}

// Re-materialises a wave-uniform pointer in SGPRs behind an opaque asm so that the compiler cannot
// hoist the (hundreds of) addresses derived from it out of the persistent tile loop -- hoisted, they
// are all live across the whole loop and spill.
template <typename T>
__device__ __forceinline__ T* launder_uniform(T* p) {
    unsigned lo = (unsigned)(reinterpret_cast<uintptr_t>(p) & 0xffffffffu);
    unsigned hi = (unsigned)(reinterpret_cast<uintptr_t>(p) >> 32);
    lo = __builtin_amdgcn_readfirstlane(lo);
    hi = __builtin_amdgcn_readfirstlane(hi);
    asm volatile("" : "+s"(lo), "+s"(hi));
    return reinterpret_cast<T*>((static_cast<uintptr_t>(hi) << 32) | lo);
}

__device__ __forceinline__ float half_sum(float v) { return v + __shfl_xor(v, 32, 64); }

}  // namespace v2
}  // namespace hn
