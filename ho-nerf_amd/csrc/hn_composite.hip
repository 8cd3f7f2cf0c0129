// SDF -> alpha (utils/renderer.py:147-161) and alpha compositing (utils/renderer.py:163-169,
// 512-524) as wavefront scans: one wave per ray, samples interleaved over the 64 lanes
// (element e = lane + 64 m, so every load/store instruction touches 64 consecutive floats),
// an inclusive product scan across the lanes per 64-sample segment and a scalar carry between
// segments.  These kernels are HBM-bound: 24 B (single) / 40 B (dual) per ray-sample.
#include "hn_common.h"

namespace hn {

__device__ __forceinline__ float sigmoid_e(float x) { return 1.f / (1.f + expf(-x)); }

__global__ void k_alpha(const float* __restrict__ sdf, const float* __restrict__ grad, const float* __restrict__ rays_d,
                        const float* __restrict__ dists, int n, int spr, float inv_s_host, float* __restrict__ alpha,
                        float* __restrict__ c_out, const float* __restrict__ inv_s_dev) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float inv_s = inv_s_dev != nullptr ? inv_s_dev[0] : inv_s_host;   // (hn_field_set_inv_s_device: a trained value that never visits the host)
    const int ray = i / spr;
    const float true_cos = rays_d[3 * ray] * grad[3 * (size_t)i] + rays_d[3 * ray + 1] * grad[3 * (size_t)i + 1] +
                           rays_d[3 * ray + 2] * grad[3 * (size_t)i + 2];
    const float iter_cos = -fmaxf(-true_cos, 0.f);     // cos_anneal_ratio = 1
    const float s = sdf[i];
    const float half = iter_cos * dists[i] * 0.5f;
    const float c = sigmoid_e((s - half) * inv_s);
    const float nx = sigmoid_e((s + half) * inv_s);
    const float a = ((c - nx) + 1e-5f) / (c + 1e-5f);
    alpha[i] = fminf(fmaxf(a, 0.f), 1.f);
    if (c_out != nullptr) c_out[i] = c;
}

__device__ __forceinline__ float wave_incl_prod(float v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float o = __shfl_up(v, off, 64);
        if (lane >= off) v *= o;
    }
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// One atomic per BLOCK for the eikonal sums: float atomics on a single address are served one at a time by the memory
// side (~12 ns each); with one per wave the 16 384 waves of a 262 144-ray launch alone took ~200 us, whatever S was.
__device__ __forceinline__ void block_atomic_add(float* dst, float wave_total) {
    __shared__ float part[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) part[wv] = wave_total;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(dst, part[0] + part[1] + part[2] + part[3]);
    __syncthreads();
}

// single field: w_k = alpha_k * c_0 * prod_{j<k} (1 - alpha_j + 1e-7)
__global__ __launch_bounds__(256) void k_composite1(const float* __restrict__ alpha, const float* __restrict__ c,
                                                    const float* __restrict__ rgb, const float* __restrict__ grad,
                                                    int n_rays, int S, float* __restrict__ color,
                                                    float* __restrict__ weights, float* __restrict__ weight_sum,
                                                    float* __restrict__ weight_max, float* __restrict__ eik_sum) {
    const int lane = threadIdx.x & 63;
    float eik_total = 0.f;   // one atomic per wave at the end (a single address takes ~12 ns per atomic)
    for (int ray = blockIdx.x * 4 + (threadIdx.x >> 6); ray < n_rays; ray += gridDim.x * 4) {
    const size_t base = (size_t)ray * S;
    float carry = c[base];      // SURVEY B-3: the first factor is c_0, not 1
    float col[3] = {0.f, 0.f, 0.f}, wsum = 0.f, wmax = -1.f, eik = 0.f;
    for (int m0 = 0; m0 < S; m0 += 64) {
        const int e = m0 + lane;
        const bool ok = e < S;
        const float a = ok ? alpha[base + e] : 0.f;
        const float fac = ok ? (1.f - a + 1e-7f) : 1.f;
        const float incl = wave_incl_prod(fac, lane);
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.f;
        const float w = a * (carry * excl);
        carry *= __shfl(incl, 63, 64);
        if (ok) {
            if (weights != nullptr) weights[base + e] = w;
            const float* r = rgb + 3 * (base + e);
            col[0] += w * r[0];
            col[1] += w * r[1];
            col[2] += w * r[2];
            wsum += w;
            wmax = fmaxf(wmax, w);
            if (grad != nullptr) {
                const float* g = grad + 3 * (base + e);
                const float nrm = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]) - 1.f;
                eik += nrm * nrm;
            }
        }
    }
    col[0] = wave_sum(col[0]);
    col[1] = wave_sum(col[1]);
    col[2] = wave_sum(col[2]);
    wsum = wave_sum(wsum);
    wmax = wave_max(wmax);
    eik = wave_sum(eik);
    if (lane == 0) {
        color[3 * ray] = col[0];
        color[3 * ray + 1] = col[1];
        color[3 * ray + 2] = col[2];
        weight_sum[ray] = wsum;
        if (weight_max != nullptr) weight_max[ray] = wmax;
    }
    eik_total += eik;
    }
    if (eik_sum != nullptr && grad != nullptr) block_atomic_add(eik_sum, eik_total);
}

// two fields: T_k = prod_{j<k} (1 - a_h + 1e-7)(1 - a_o + 1e-7); w_h = a_h T, w_o = a_o T
__global__ __launch_bounds__(256) void k_composite2(const float* __restrict__ ah, const float* __restrict__ rgbh,
                                                    const float* __restrict__ gh, const float* __restrict__ ao,
                                                    const float* __restrict__ rgbo, const float* __restrict__ go,
                                                    int n_rays, int S, float* __restrict__ color,
                                                    float* __restrict__ weight_sum, float* __restrict__ w_hand,
                                                    float* __restrict__ w_obj, float* __restrict__ eik_sum, float eik_scale) {
    const int lane = threadIdx.x & 63;
    float eh_total = 0.f, eo_total = 0.f;
    for (int ray = blockIdx.x * 4 + (threadIdx.x >> 6); ray < n_rays; ray += gridDim.x * 4) {
    const size_t base = (size_t)ray * S;
    float carry = 1.f;
    float colh[3] = {0.f, 0.f, 0.f}, colo[3] = {0.f, 0.f, 0.f}, wsh = 0.f, wso = 0.f, eh = 0.f, eo = 0.f;
    for (int m0 = 0; m0 < S; m0 += 64) {
        const int e = m0 + lane;
        const bool ok = e < S;
        const float a1 = ok ? ah[base + e] : 0.f;
        const float a2 = ok ? ao[base + e] : 0.f;
        const float fac = ok ? (1.f - a1 + 1e-7f) * (1.f - a2 + 1e-7f) : 1.f;
        const float incl = wave_incl_prod(fac, lane);
        float excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.f;
        const float T = carry * excl;
        carry *= __shfl(incl, 63, 64);
        if (ok) {
            const float w1 = a1 * T, w2 = a2 * T;
            if (w_hand != nullptr) w_hand[base + e] = w1;
            if (w_obj != nullptr) w_obj[base + e] = w2;
            const float* r1 = rgbh + 3 * (base + e);
            const float* r2 = rgbo + 3 * (base + e);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                colh[k] += w1 * r1[k];
                colo[k] += w2 * r2[k];
            }
            wsh += w1;
            wso += w2;
            if (gh != nullptr) {
                const float* g = gh + 3 * (base + e);
                const float nrm = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]) - 1.f;
                eh += nrm * nrm;
            }
            if (go != nullptr) {
                const float* g = go + 3 * (base + e);
                const float nrm = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]) - 1.f;
                eo += nrm * nrm;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        colh[k] = wave_sum(colh[k]);
        colo[k] = wave_sum(colo[k]);
    }
    wsh = wave_sum(wsh);
    wso = wave_sum(wso);
    eh = wave_sum(eh);
    eo = wave_sum(eo);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) color[3 * ray + k] = colh[k] + colo[k];
        weight_sum[ray] = wsh + wso;
    }
    eh_total += eh;
    eo_total += eo;
    }
    if (eik_sum != nullptr) {
        if (gh != nullptr) block_atomic_add(eik_sum, eh_total * eik_scale);
        if (go != nullptr) block_atomic_add(eik_sum + 1, eo_total * eik_scale);
    }
}

// ---- S a multiple of 64: 16 lanes per ray, every lane owns S/16 consecutive samples -----------------------------
// A wave takes 4 consecutive rays; lane (row r = lane >> 4, l = lane & 15) owns samples l*CPS .. l*CPS + CPS-1 of ray
// 4 g + r, so a wave reads 4*S consecutive floats of every per-sample array with 16-byte loads.  The transmittance is a
// sequential product inside the lane, an exclusive product scan over the 16 lanes of the row (DPP row_shr: VALU only,
// where __shfl_up goes through the LDS crossbar) and the per-ray sums are row_ror butterflies.
template <int CTRL>
__device__ __forceinline__ float dpp(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL,
                                                                  0xf, 0xf, false));
}
// inclusive product over the 16 lanes of a row (row_shr:n = 0x110 + n; lanes without a source keep `old` = 1)
__device__ __forceinline__ float row_incl_prod(float v) {
    v *= dpp<0x111>(1.f, v);
    v *= dpp<0x112>(1.f, v);
    v *= dpp<0x114>(1.f, v);
    v *= dpp<0x118>(1.f, v);
    return v;
}
// sum / max over the row, result in every lane (row_ror:n = 0x120 + n)
__device__ __forceinline__ float row_sum(float v) {
    v += dpp<0x128>(0.f, v);
    v += dpp<0x124>(0.f, v);
    v += dpp<0x122>(0.f, v);
    v += dpp<0x121>(0.f, v);
    return v;
}
__device__ __forceinline__ float row_max(float v) {
    v = fmaxf(v, dpp<0x128>(0.f, v));
    v = fmaxf(v, dpp<0x124>(0.f, v));
    v = fmaxf(v, dpp<0x122>(0.f, v));
    v = fmaxf(v, dpp<0x121>(0.f, v));
    return v;
}
// loads CPS consecutive floats / CPS consecutive float3 of this lane (16-byte accesses)
template <int CPS>
__device__ __forceinline__ void load_run(const float* __restrict__ p, float (&v)[CPS]) {
#pragma unroll
    for (int q = 0; q < CPS / 4; ++q) {
        const float4 t = reinterpret_cast<const float4*>(p)[q];
        v[4 * q] = t.x;
        v[4 * q + 1] = t.y;
        v[4 * q + 2] = t.z;
        v[4 * q + 3] = t.w;
    }
}
template <int CPS>
__device__ __forceinline__ void store_run(float* __restrict__ p, const float (&v)[CPS]) {
#pragma unroll
    for (int q = 0; q < CPS / 4; ++q) reinterpret_cast<float4*>(p)[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}
__device__ __forceinline__ float wave_sum_rows(float v) {   // v is already a row sum (equal within a row)
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// sum / max over a group of LPR (8 or 16) consecutive lanes, result in every lane of the group: xor butterflies as DPP
// (quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140): VALU only
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
    v += dpp<0xB1>(0.f, v);
    v += dpp<0x4E>(0.f, v);
    v += dpp<0x141>(0.f, v);
    if (LPR == 16) v += dpp<0x140>(0.f, v);
    return v;
}
template <int LPR>
__device__ __forceinline__ float group_max(float v) {
    v = fmaxf(v, dpp<0xB1>(0.f, v));
    v = fmaxf(v, dpp<0x4E>(0.f, v));
    v = fmaxf(v, dpp<0x141>(0.f, v));
    if (LPR == 16) v = fmaxf(v, dpp<0x140>(0.f, v));
    return v;
}
// exclusive product scan over the group (lane l of the group gets the product of lanes 0 .. l-1; lane 0 gets 1)
template <int LPR>
__device__ __forceinline__ float group_excl_prod(float v, int l) {
    float t;
    t = dpp<0x111>(1.f, v);
    v *= (LPR == 16 || l >= 1) ? t : 1.f;
    t = dpp<0x112>(1.f, v);
    v *= (LPR == 16 || l >= 2) ? t : 1.f;
    t = dpp<0x114>(1.f, v);
    v *= (LPR == 16 || l >= 4) ? t : 1.f;
    if (LPR == 16) v *= dpp<0x118>(1.f, v);
    t = dpp<0x111>(1.f, v);          // shift the inclusive scan by one lane
    return l >= 1 ? t : 1.f;
}
template <int N>
__device__ __forceinline__ float eik_run(const float (&x)[3 * N]) {
    float e = 0.f;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float nrm = sqrtf(x[3 * i] * x[3 * i] + x[3 * i + 1] * x[3 * i + 1] + x[3 * i + 2] * x[3 * i + 2]) - 1.f;
        e += nrm * nrm;
    }
    return e;
}

// LPR lanes per ray, CPS consecutive samples per lane (S = LPR * CPS), 64 / LPR rays per wave.  Every load of an
// iteration is issued before the first use (16 bytes per lane each; 7 CPS / 4 of them per lane are in flight), the
// transmittance is a sequential product inside the lane, an exclusive product scan over the group and group sums, all
// DPP.  (Round 1 loaded rgb / grad after the scan: 0.39 - 0.57 of the HBM peak; the loads first: see profiles/r02.)
template <int CPS, int LPR>
__global__ __launch_bounds__(256) void k_composite1_rows(const float* __restrict__ alpha, const float* __restrict__ c,
                                                         const float* __restrict__ rgb, const float* __restrict__ grad,
                                                         int n_rays, float* __restrict__ color, float* __restrict__ weights,
                                                         float* __restrict__ weight_sum, float* __restrict__ weight_max,
                                                         float* __restrict__ eik_sum) {
    constexpr int S = LPR * CPS, RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, l = lane & (LPR - 1), row = lane / LPR;
    float eik_total = 0.f;
    for (int g = blockIdx.x * 4 + (threadIdx.x >> 6); g * RPW < n_rays; g += gridDim.x * 4) {
        const int ray = g * RPW + row;
        const bool ok = ray < n_rays;
        const size_t rbase = (size_t)(ok ? ray : n_rays - 1) * S;
        const size_t base = rbase + l * CPS;
        float a[CPS], w[CPS], x[3 * CPS], y[3 * CPS];
        load_run<CPS>(alpha + base, a);
        load_run<3 * CPS>(rgb + 3 * base, x);
        if (grad != nullptr) load_run<3 * CPS>(grad + 3 * base, y);
        const float c0 = c[rbase];   // SURVEY B-3: the first factor is c_0, not 1
        float P = 1.f;
#pragma unroll
        for (int i = 0; i < CPS; ++i) P *= 1.f - a[i] + 1e-7f;
        float T = c0 * group_excl_prod<LPR>(P, l);
        float wsum = 0.f, wmax = -1.f;
        float col[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < CPS; ++i) {
            w[i] = a[i] * T;
            T *= 1.f - a[i] + 1e-7f;
            wsum += w[i];
            wmax = fmaxf(wmax, w[i]);
            col[0] += w[i] * x[3 * i];
            col[1] += w[i] * x[3 * i + 1];
            col[2] += w[i] * x[3 * i + 2];
        }
        if (ok && weights != nullptr) store_run<CPS>(weights + base, w);
        const float eik = grad != nullptr ? eik_run<CPS>(y) : 0.f;
        col[0] = group_sum<LPR>(col[0]);
        col[1] = group_sum<LPR>(col[1]);
        col[2] = group_sum<LPR>(col[2]);
        wsum = group_sum<LPR>(wsum);
        wmax = group_max<LPR>(wmax);
        if (ok && l == 0) {
            color[3 * ray] = col[0];
            color[3 * ray + 1] = col[1];
            color[3 * ray + 2] = col[2];
            weight_sum[ray] = wsum;
            if (weight_max != nullptr) weight_max[ray] = wmax;
        }
        eik_total += ok ? eik : 0.f;
    }
    if (eik_sum != nullptr && grad != nullptr) block_atomic_add(eik_sum, wave_sum(eik_total));
}

template <int CPS, int LPR>
__global__ __launch_bounds__(256) void k_composite2_rows(const float* __restrict__ ah, const float* __restrict__ rgbh,
                                                         const float* __restrict__ gh, const float* __restrict__ ao,
                                                         const float* __restrict__ rgbo, const float* __restrict__ go,
                                                         int n_rays, float* __restrict__ color, float* __restrict__ weight_sum,
                                                         float* __restrict__ w_hand, float* __restrict__ w_obj,
                                                         float* __restrict__ eik_sum, float eik_scale) {
    constexpr int S = LPR * CPS, RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, l = lane & (LPR - 1), row = lane / LPR;
    float eh_total = 0.f, eo_total = 0.f;
    for (int g = blockIdx.x * 4 + (threadIdx.x >> 6); g * RPW < n_rays; g += gridDim.x * 4) {
        const int ray = g * RPW + row;
        const bool ok = ray < n_rays;
        const size_t base = (size_t)(ok ? ray : n_rays - 1) * S + l * CPS;
        float a1[CPS], a2[CPS], w1[CPS], w2[CPS], x1[3 * CPS], x2[3 * CPS];
        load_run<CPS>(ah + base, a1);
        load_run<CPS>(ao + base, a2);
        load_run<3 * CPS>(rgbh + 3 * base, x1);
        load_run<3 * CPS>(rgbo + 3 * base, x2);
        float P = 1.f;
#pragma unroll
        for (int i = 0; i < CPS; ++i) P *= (1.f - a1[i] + 1e-7f) * (1.f - a2[i] + 1e-7f);
        float T = group_excl_prod<LPR>(P, l);
        float ws = 0.f;
        float col[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < CPS; ++i) {
            w1[i] = a1[i] * T;
            w2[i] = a2[i] * T;
            T *= (1.f - a1[i] + 1e-7f) * (1.f - a2[i] + 1e-7f);
            ws += w1[i] + w2[i];
            col[0] += w1[i] * x1[3 * i] + w2[i] * x2[3 * i];
            col[1] += w1[i] * x1[3 * i + 1] + w2[i] * x2[3 * i + 1];
            col[2] += w1[i] * x1[3 * i + 2] + w2[i] * x2[3 * i + 2];
        }
        if (ok && w_hand != nullptr) store_run<CPS>(w_hand + base, w1);
        if (ok && w_obj != nullptr) store_run<CPS>(w_obj + base, w2);
        // the gradients (eikonal term) re-use the colour registers
        float eh = 0.f, eo = 0.f;
        if (gh != nullptr) load_run<3 * CPS>(gh + 3 * base, x1);
        if (go != nullptr) load_run<3 * CPS>(go + 3 * base, x2);
        if (gh != nullptr) eh = eik_run<CPS>(x1);
        if (go != nullptr) eo = eik_run<CPS>(x2);
        col[0] = group_sum<LPR>(col[0]);
        col[1] = group_sum<LPR>(col[1]);
        col[2] = group_sum<LPR>(col[2]);
        ws = group_sum<LPR>(ws);
        if (ok && l == 0) {
            color[3 * ray] = col[0];
            color[3 * ray + 1] = col[1];
            color[3 * ray + 2] = col[2];
            weight_sum[ray] = ws;
        }
        eh_total += ok ? eh : 0.f;
        eo_total += ok ? eo : 0.f;
    }
    if (eik_sum != nullptr) {
        if (gh != nullptr) block_atomic_add(eik_sum, wave_sum(eh_total) * eik_scale);
        if (go != nullptr) block_atomic_add(eik_sum + 1, wave_sum(eo_total) * eik_scale);
    }
}

// ---- adjoints (pose fitting back-propagates through these stages: fitting_single.py:289-291) ---------
// d/d(sdf, grad, rays_d) of k_alpha.  One thread per sample; the per-ray direction gradient is accumulated
// with atomics (S adders per ray; the fitting configs have ~200 rays).  dists carry no gradient: they come
// from depths the reference samples under no_grad (utils/renderer.py:215, 461).
__global__ void k_alpha_bwd(const float* __restrict__ sdf, const float* __restrict__ grad, const float* __restrict__ rays_d,
                            const float* __restrict__ dists, const float* __restrict__ g_alpha,
                            const float* __restrict__ g_c, int n, int spr, float inv_s_host, float* __restrict__ g_sdf,
                            float* __restrict__ g_grad, float* __restrict__ g_rays_d, const float* __restrict__ inv_s_dev) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;   // (the wave-level reduction below checks that all 64 lanes are present)
    const float inv_s = inv_s_dev != nullptr ? inv_s_dev[0] : inv_s_host;
    const int ray = i / spr;
    const float d0 = rays_d[3 * ray], d1 = rays_d[3 * ray + 1], d2 = rays_d[3 * ray + 2];
    const float q0 = grad[3 * (size_t)i], q1 = grad[3 * (size_t)i + 1], q2 = grad[3 * (size_t)i + 2];
    const float tc = d0 * q0 + d1 * q1 + d2 * q2;
    const float ic = -fmaxf(-tc, 0.f);
    const float s = sdf[i], dist = dists[i];
    const float half = ic * dist * 0.5f;
    const float c = sigmoid_e((s - half) * inv_s);
    const float nx = sigmoid_e((s + half) * inv_s);
    const float A = (c - nx) + 1e-5f, B = c + 1e-5f;
    const float a_raw = A / B;
    const float ga = (a_raw > 0.f && a_raw < 1.f) ? g_alpha[i] : 0.f;   // clip(0, 1)
    // a = A / B: da/dc = (B - A) / B^2, da/dnx = -1 / B
    float gc = ga * (B - A) / (B * B) + (g_c != nullptr ? g_c[i] : 0.f);
    float gnx = -ga / B;
    const float gx1 = gc * inv_s * c * (1.f - c);      // x1 = s - half
    const float gx2 = gnx * inv_s * nx * (1.f - nx);   // x2 = s + half
    g_sdf[i] = gx1 + gx2;
    const float ghalf = gx2 - gx1;
    const float gtc = (tc < 0.f) ? ghalf * dist * 0.5f : 0.f;   // ic = min(tc, 0)
    g_grad[3 * (size_t)i] = gtc * d0;
    g_grad[3 * (size_t)i + 1] = gtc * d1;
    g_grad[3 * (size_t)i + 2] = gtc * d2;
    if (g_rays_d != nullptr) {
        // the 64 samples of a wave usually belong to one ray (spr a multiple of 64: 3 atomics per wave instead of 192);
        // S adders on one address were the kernel's whole time at the fitting sizes (78 of 80 us for 196 x 192 samples)
        const int ray0 = __shfl(ray, 0, 64);
        const bool whole = __all(ray == ray0) && __popcll(__ballot(1)) == 64;
        if (whole) {
            const float r0 = wave_sum(gtc * q0), r1 = wave_sum(gtc * q1), r2 = wave_sum(gtc * q2);
            if ((threadIdx.x & 63) == 0) {
                atomicAdd(g_rays_d + 3 * ray, r0);
                atomicAdd(g_rays_d + 3 * ray + 1, r1);
                atomicAdd(g_rays_d + 3 * ray + 2, r2);
            }
        } else {
            atomicAdd(g_rays_d + 3 * ray, gtc * q0);
            atomicAdd(g_rays_d + 3 * ray + 1, gtc * q1);
            atomicAdd(g_rays_d + 3 * ray + 2, gtc * q2);
        }
    }
}

// Adjoint of the transmittance products without dividing by a factor that may be ~1e-7: with
// L = sum_k T_k s_k (s_k = what multiplies T_k) and T_k = seed * prod_{j<k} f_j,
//   dL/df_k = T_k P_k,   P_k = s_{k+1} + f_{k+1} P_{k+1}  (P_{S-1} = 0),   dL/dseed = (s_0 + f_0 P_0) = P_{-1}.
// One thread per ray, sequential over its S samples (latency-bound at the sizes back-propagation runs at).
__global__ void k_composite1_bwd(const float* __restrict__ alpha, const float* __restrict__ c, const float* __restrict__ rgb,
                                 const float* __restrict__ g_color, const float* __restrict__ g_wsum, int n_rays, int S,
                                 float* __restrict__ g_alpha, float* __restrict__ g_c, float* __restrict__ g_rgb) {
    const int ray = blockIdx.x * blockDim.x + threadIdx.x;
    if (ray >= n_rays) return;
    const size_t base = (size_t)ray * S;
    const float gC0 = g_color[3 * ray], gC1 = g_color[3 * ray + 1], gC2 = g_color[3 * ray + 2];
    const float gW = g_wsum != nullptr ? g_wsum[ray] : 0.f;
    // forward transmittances into g_alpha (scratch), seeded with c_0 (SURVEY B-3)
    float T = c[base];
    for (int k = 0; k < S; ++k) {
        const float a = alpha[base + k];
        g_alpha[base + k] = T;
        const float w = a * T;
        g_rgb[3 * (base + k)] = w * gC0;
        g_rgb[3 * (base + k) + 1] = w * gC1;
        g_rgb[3 * (base + k) + 2] = w * gC2;
        T *= (1.f - a + 1e-7f);
    }
    float P = 0.f;
    for (int k = S - 1; k >= 0; --k) {
        const float a = alpha[base + k];
        const float* r = rgb + 3 * (base + k);
        const float u = gC0 * r[0] + gC1 * r[1] + gC2 * r[2] + gW;   // dL/dw_k
        const float Tk = g_alpha[base + k];
        g_alpha[base + k] = Tk * u - Tk * P;        // w = a T ; f = 1 - a + 1e-7
        g_c[base + k] = 0.f;
        P = a * u + (1.f - a + 1e-7f) * P;
    }
    g_c[base] = P;   // dL/dseed: T_k is linear in the seed c_0
}

__global__ void k_composite2_bwd(const float* __restrict__ ah, const float* __restrict__ rgbh, const float* __restrict__ ao,
                                 const float* __restrict__ rgbo, const float* __restrict__ g_color,
                                 const float* __restrict__ g_wsum, int n_rays, int S, float* __restrict__ g_ah,
                                 float* __restrict__ g_rgbh, float* __restrict__ g_ao, float* __restrict__ g_rgbo) {
    const int ray = blockIdx.x * blockDim.x + threadIdx.x;
    if (ray >= n_rays) return;
    const size_t base = (size_t)ray * S;
    const float gC0 = g_color[3 * ray], gC1 = g_color[3 * ray + 1], gC2 = g_color[3 * ray + 2];
    const float gW = g_wsum != nullptr ? g_wsum[ray] : 0.f;
    float T = 1.f;
    for (int k = 0; k < S; ++k) {
        const float a1 = ah[base + k], a2 = ao[base + k];
        g_ah[base + k] = T;
        const float w1 = a1 * T, w2 = a2 * T;
        g_rgbh[3 * (base + k)] = w1 * gC0;
        g_rgbh[3 * (base + k) + 1] = w1 * gC1;
        g_rgbh[3 * (base + k) + 2] = w1 * gC2;
        g_rgbo[3 * (base + k)] = w2 * gC0;
        g_rgbo[3 * (base + k) + 1] = w2 * gC1;
        g_rgbo[3 * (base + k) + 2] = w2 * gC2;
        T *= (1.f - a1 + 1e-7f) * (1.f - a2 + 1e-7f);
    }
    float P = 0.f;
    for (int k = S - 1; k >= 0; --k) {
        const float a1 = ah[base + k], a2 = ao[base + k];
        const float* r1 = rgbh + 3 * (base + k);
        const float* r2 = rgbo + 3 * (base + k);
        const float u1 = gC0 * r1[0] + gC1 * r1[1] + gC2 * r1[2] + gW;
        const float u2 = gC0 * r2[0] + gC1 * r2[1] + gC2 * r2[2] + gW;
        const float Tk = g_ah[base + k];
        const float f1 = 1.f - a1 + 1e-7f, f2 = 1.f - a2 + 1e-7f;
        g_ah[base + k] = Tk * u1 - Tk * P * f2;     // df/da1 = -f2
        g_ao[base + k] = Tk * u2 - Tk * P * f1;     // df/da2 = -f1
        P = (a1 * u1 + a2 * u2) + f1 * f2 * P;
    }
}

// The same adjoint with one WAVE per ray (S <= 64 * CPL): lane l owns the CPL consecutive samples CPL l .. CPL l + CPL - 1,
// the transmittance is an exclusive product scan over the lanes, and the suffix recurrence P_k = c_k + m_k P_{k+1} is a
// suffix scan of the affine maps x -> c + m x (composition (m1, c1) o (m2, c2) = (m1 m2, c1 + m1 c2)).  The one-thread-per-ray
// form above walks 192 dependent, uncoalesced rows per ray: 152 us at the fitting size (196 rays), against ~6 us here.
template <int CPL>
__global__ __launch_bounds__(256) void k_composite2_bwd_wave(const float* __restrict__ ah, const float* __restrict__ rgbh,
                                                             const float* __restrict__ ao, const float* __restrict__ rgbo,
                                                             const float* __restrict__ g_color, const float* __restrict__ g_wsum,
                                                             int n_rays, int S, float* __restrict__ g_ah, float* __restrict__ g_rgbh,
                                                             float* __restrict__ g_ao, float* __restrict__ g_rgbo) {
    const int lane = threadIdx.x & 63;
    const int ray = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const size_t base = (size_t)ray * S;
    const float gC0 = g_color[3 * ray], gC1 = g_color[3 * ray + 1], gC2 = g_color[3 * ray + 2];
    const float gW = g_wsum != nullptr ? g_wsum[ray] : 0.f;
    float a1[CPL], a2[CPL], u1[CPL], u2[CPL], f1[CPL], f2[CPL];
    float Fl = 1.f;   // product of this lane's factors
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        const int k = lane * CPL + j;
        const bool in = k < S;
        a1[j] = in ? ah[base + k] : 0.f;
        a2[j] = in ? ao[base + k] : 0.f;
        u1[j] = u2[j] = 0.f;
        if (in) {
            const float* r1 = rgbh + 3 * (base + k);
            const float* r2 = rgbo + 3 * (base + k);
            u1[j] = gC0 * r1[0] + gC1 * r1[1] + gC2 * r1[2] + gW;
            u2[j] = gC0 * r2[0] + gC1 * r2[1] + gC2 * r2[2] + gW;
        }
        f1[j] = in ? 1.f - a1[j] + 1e-7f : 1.f;
        f2[j] = in ? 1.f - a2[j] + 1e-7f : 1.f;
        Fl *= f1[j] * f2[j];
    }
    // exclusive product scan over the lanes -> transmittance in front of this lane's first sample
    float incl = Fl;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float o = __shfl_up(incl, off, 64);
        if (lane >= off) incl *= o;
    }
    float T = __shfl_up(incl, 1, 64);
    if (lane == 0) T = 1.f;
    float Tk[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        Tk[j] = T;
        T *= f1[j] * f2[j];
    }
    // this lane's composite map x -> C + M x (its samples applied last to first), then the suffix scan over the lanes
    float M = 1.f, C = 0.f;
#pragma unroll
    for (int j = CPL - 1; j >= 0; --j) {
        const float m = f1[j] * f2[j], c = a1[j] * u1[j] + a2[j] * u2[j];
        C = c + m * C;
        M = m * M;
    }
    float sm = M, sc = C;   // inclusive: lanes l .. 63
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float om = __shfl_down(sm, off, 64), oc = __shfl_down(sc, off, 64);
        if (lane + off < 64) {
            sc = sc + sm * oc;
            sm = sm * om;
        }
    }
    float P = __shfl_down(sc, 1, 64);   // the recurrence's value behind this lane's last sample (maps applied to 0)
    if (lane == 63) P = 0.f;
#pragma unroll
    for (int j = CPL - 1; j >= 0; --j) {
        const int k = lane * CPL + j;
        if (k < S) {
            g_ah[base + k] = Tk[j] * u1[j] - Tk[j] * P * f2[j];
            g_ao[base + k] = Tk[j] * u2[j] - Tk[j] * P * f1[j];
            const float w1 = a1[j] * Tk[j], w2 = a2[j] * Tk[j];
            g_rgbh[3 * (base + k)] = w1 * gC0;
            g_rgbh[3 * (base + k) + 1] = w1 * gC1;
            g_rgbh[3 * (base + k) + 2] = w1 * gC2;
            g_rgbo[3 * (base + k)] = w2 * gC0;
            g_rgbo[3 * (base + k) + 1] = w2 * gC1;
            g_rgbo[3 * (base + k) + 2] = w2 * gC2;
        }
        P = (a1[j] * u1[j] + a2[j] * u2[j]) + f1[j] * f2[j] * P;
    }
}

// The one-field adjoint (k_composite1_bwd) with one wave per ray, as k_composite2_bwd_wave: T_k = c_0 prod_{j<k} f_j is the seed times an
// exclusive product scan, P_k a suffix scan of affine maps; g_c = (P_{-1}, 0, 0, ..).  The thread-per-ray form took 67 us for the 441 x 128
// samples of a training iteration (128 dependent, uncoalesced rows per ray), this one ~6.
template <int CPL>
__global__ __launch_bounds__(256) void k_composite1_bwd_wave(const float* __restrict__ alpha, const float* __restrict__ c, const float* __restrict__ rgb,
                                                             const float* __restrict__ g_color, const float* __restrict__ g_wsum, int n_rays, int S,
                                                             float* __restrict__ g_alpha, float* __restrict__ g_c, float* __restrict__ g_rgb) {
    const int lane = threadIdx.x & 63;
    const int ray = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const size_t base = (size_t)ray * S;
    const float gC0 = g_color[3 * ray], gC1 = g_color[3 * ray + 1], gC2 = g_color[3 * ray + 2];
    const float gW = g_wsum != nullptr ? g_wsum[ray] : 0.f;
    const float seed = c[base];
    float a[CPL], u[CPL], f[CPL];
    float Fl = 1.f;   // product of this lane's factors
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        const int k = lane * CPL + j;
        const bool in = k < S;
        a[j] = in ? alpha[base + k] : 0.f;
        u[j] = 0.f;
        if (in) {
            const float* r = rgb + 3 * (base + k);
            u[j] = gC0 * r[0] + gC1 * r[1] + gC2 * r[2] + gW;   // dL/dw_k
        }
        f[j] = in ? 1.f - a[j] + 1e-7f : 1.f;
        Fl *= f[j];
    }
    float incl = Fl;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float o = __shfl_up(incl, off, 64);
        if (lane >= off) incl *= o;
    }
    float T = __shfl_up(incl, 1, 64);
    T = lane == 0 ? seed : seed * T;   // the transmittance in front of this lane's first sample (seeded with c_0: SURVEY B-3)
    float Tk[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        Tk[j] = T;
        T *= f[j];
    }
    float M = 1.f, C = 0.f;   // this lane's composite map x -> C + M x (its samples applied last to first)
#pragma unroll
    for (int j = CPL - 1; j >= 0; --j) {
        C = a[j] * u[j] + f[j] * C;
        M = f[j] * M;
    }
    float sm = M, sc = C;   // inclusive suffix: lanes l .. 63
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float om = __shfl_down(sm, off, 64), oc = __shfl_down(sc, off, 64);
        if (lane + off < 64) {
            sc = sc + sm * oc;
            sm = sm * om;
        }
    }
    float P = __shfl_down(sc, 1, 64);   // the recurrence's value behind this lane's last sample
    if (lane == 63) P = 0.f;
#pragma unroll
    for (int j = CPL - 1; j >= 0; --j) {
        const int k = lane * CPL + j;
        if (k < S) {
            g_alpha[base + k] = Tk[j] * u[j] - Tk[j] * P;
            const float w = a[j] * Tk[j];
            g_rgb[3 * (base + k)] = w * gC0;
            g_rgb[3 * (base + k) + 1] = w * gC1;
            g_rgb[3 * (base + k) + 2] = w * gC2;
        }
        P = a[j] * u[j] + f[j] * P;
        if (k < S) g_c[base + k] = k == 0 ? P : 0.f;   // dL/dseed = P_{-1}: T_k is linear in the seed c_0
    }
}

// 4 rays per block; at most 8 blocks per CU, grid-stride beyond that
static int composite_grid(int n_rays) {
    const int blocks = (n_rays + 3) / 4;
    return blocks < 2048 ? blocks : 2048;
}

int alpha(const float* sdf, const float* grad, const float* rays_d, const float* dists, int n, int spr, float inv_s,
          float* alpha_out, float* c, hipStream_t s, const float* inv_s_dev) {
    HN_REQUIRE(spr > 0, "samples_per_ray must be positive");
    if (n == 0) return HN_OK;
    hipLaunchKernelGGL(k_alpha, dim3((n + 255) / 256), dim3(256), 0, s, sdf, grad, rays_d, dists, n, spr, inv_s,
                       alpha_out, c, inv_s_dev);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

int composite1(const float* alpha_in, const float* c, const float* rgb, const float* grad, int n_rays, int S,
               float* color, float* weights, float* weight_sum, float* weight_max, float* eik_sum, hipStream_t s) {
    HN_REQUIRE(S >= 1, "S must be positive");
    if (n_rays == 0) return HN_OK;
    // row kernels: LPR lanes per ray (64 / LPR rays per wave), 4 waves per block
    auto rgrid = [&](int lpr) {
        const int groups = (n_rays + 64 / lpr - 1) / (64 / lpr);
        return dim3((groups + 3) / 4 < 2048 ? (groups + 3) / 4 : 2048);
    };
    if (S == 64)
        hipLaunchKernelGGL((k_composite1_rows<4, 16>), rgrid(16), dim3(256), 0, s, alpha_in, c, rgb, grad, n_rays, color, weights,
                           weight_sum, weight_max, eik_sum);
    else if (S == 128)
        hipLaunchKernelGGL((k_composite1_rows<8, 16>), rgrid(16), dim3(256), 0, s, alpha_in, c, rgb, grad, n_rays, color, weights,
                           weight_sum, weight_max, eik_sum);
    else if (S == 192)
        hipLaunchKernelGGL((k_composite1_rows<12, 16>), rgrid(16), dim3(256), 0, s, alpha_in, c, rgb, grad, n_rays, color, weights,
                           weight_sum, weight_max, eik_sum);
    else if (S == 32)
        hipLaunchKernelGGL((k_composite1_rows<4, 8>), rgrid(8), dim3(256), 0, s, alpha_in, c, rgb, grad, n_rays, color, weights,
                           weight_sum, weight_max, eik_sum);
    else
        hipLaunchKernelGGL(k_composite1, dim3(composite_grid(n_rays)), dim3(256), 0, s, alpha_in, c, rgb, grad, n_rays, S, color,
                           weights, weight_sum, weight_max, eik_sum);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

// eik_scale: what the two eikonal sums are multiplied by as they are accumulated (1: the plain sums of hn_composite2; the
// two-field render passes 1 / (rays x S), the mean of utils/renderer.py:417, instead of a scaling launch behind this one)
int composite2(const float* ah, const float* rgbh, const float* gh, const float* ao, const float* rgbo, const float* go,
               int n_rays, int S, float* color, float* weight_sum, float* w_hand, float* w_obj, float* eik_sum,
               hipStream_t s, float eik_scale) {
    HN_REQUIRE(S >= 1, "S must be positive");
    if (n_rays == 0) return HN_OK;
    auto rgrid = [&](int lpr) {
        const int groups = (n_rays + 64 / lpr - 1) / (64 / lpr);
        return dim3((groups + 3) / 4 < 2048 ? (groups + 3) / 4 : 2048);
    };
    if (S == 64)
        hipLaunchKernelGGL((k_composite2_rows<4, 16>), rgrid(16), dim3(256), 0, s, ah, rgbh, gh, ao, rgbo, go, n_rays, color, weight_sum,
                           w_hand, w_obj, eik_sum, eik_scale);
    else if (S == 128)
        hipLaunchKernelGGL((k_composite2_rows<8, 16>), rgrid(16), dim3(256), 0, s, ah, rgbh, gh, ao, rgbo, go, n_rays, color, weight_sum,
                           w_hand, w_obj, eik_sum, eik_scale);
    else if (S == 192)
        hipLaunchKernelGGL((k_composite2_rows<12, 16>), rgrid(16), dim3(256), 0, s, ah, rgbh, gh, ao, rgbo, go, n_rays, color, weight_sum,
                           w_hand, w_obj, eik_sum, eik_scale);
    else
        hipLaunchKernelGGL(k_composite2, dim3(composite_grid(n_rays)), dim3(256), 0, s, ah, rgbh, gh, ao, rgbo, go, n_rays, S,
                           color, weight_sum, w_hand, w_obj, eik_sum, eik_scale);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

int alpha_bwd(const float* sdf, const float* grad, const float* rays_d, const float* dists, const float* g_alpha,
              const float* g_c, int n, int spr, float inv_s, float* g_sdf, float* g_grad, float* g_rays_d, hipStream_t s,
              bool g_rays_d_zeroed, const float* inv_s_dev) {
    HN_REQUIRE(spr > 0, "samples_per_ray must be positive");
    if (n == 0) return HN_OK;
    if (g_rays_d != nullptr && !g_rays_d_zeroed) HN_CHECK_HIP(hipMemsetAsync(g_rays_d, 0, (size_t)(n / spr) * 3 * sizeof(float), s));
    hipLaunchKernelGGL(k_alpha_bwd, dim3((n + 255) / 256), dim3(256), 0, s, sdf, grad, rays_d, dists, g_alpha, g_c, n, spr,
                       inv_s, g_sdf, g_grad, g_rays_d, inv_s_dev);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

// The preamble of a field's adjoint in the two-field backward pass as ONE launch (it was alpha_bwd + k_upstream [+ the gather
// onto the compacted sample list] [+ a zero fill] [+ a sample_points launch for the dists]: a step is a chain of dependent
// launches, ~5 us each whatever they do).  Per sample: k_alpha_bwd's values, with the section length taken from the depths
// (utils/renderer.py:119-120: z_{k+1} - z_k, the last one sample_dist) instead of a dists array;
//   gs = d/d sdf through alpha + the caller's direct gradient on the sdf output,
//   gg = d/d gradient through alpha + the caller's direct gradient + the eikonal term (k_upstream, hn_api.hip);
// with `pos` (hn_field_set_compaction) the live samples' (gs, gg, g_rgb) also go to their slots of the compact list and the
// stand-in's slot n_dev[0] - 1 is zeroed (it stands for samples that contribute nothing); `zero`: buffers the adjoint kernel
// behind this launch accumulates into.
struct AlphaUpZero {
    float* p[4];
    int n[4];
};
__global__ __launch_bounds__(256) void k_alpha_bwd_up(const float* __restrict__ sdf, const float* __restrict__ grad, const float* __restrict__ rays_d,
                                                      const float* __restrict__ z, const float* __restrict__ g_alpha, int n, int spr, float sample_dist,
                                                      float inv_s, const float* __restrict__ g_sdf_out, const float* __restrict__ g_grad_out,
                                                      const float* __restrict__ g_eik, float* __restrict__ gs, float* __restrict__ gg,
                                                      float* __restrict__ g_rays_d, const int* __restrict__ pos, const int* __restrict__ n_dev,
                                                      const float* __restrict__ g_rgb, float* __restrict__ gs_c, float* __restrict__ gg_c,
                                                      float* __restrict__ gr_c, AlphaUpZero zero, float* __restrict__ g_rays_d_s,
                                                      const int* __restrict__ seg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int b = 0; b < 4; ++b)
        if (zero.p[b] != nullptr && i < zero.n[b]) zero.p[b][i] = 0.f;
    if (pos != nullptr && seg != nullptr) {
        // frame-aligned compact list (hn_api.hip, k_hand_compact_write): the pads behind a frame's live samples get zero upstream gradients
        const int af = n_dev[2];
        const int f = i >> 7;
        if (f + 1 < af) {
            const int slot = seg[f] + seg[af + 1 + f] + (i & 127);
            if (slot < seg[f + 1]) {
                gs_c[slot] = 0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    gg_c[3 * (size_t)slot + c] = 0.f;
                    gr_c[3 * (size_t)slot + c] = 0.f;
                }
            }
        }
    }
    if (pos != nullptr && i == 0) {
        const int M = n_dev[0] - 1;
        gs_c[M] = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            gg_c[3 * (size_t)M + c] = 0.f;
            gr_c[3 * (size_t)M + c] = 0.f;
        }
    }
    if (i >= n) return;
    const int ray = i / spr, k = i - ray * spr;
    const float d0 = rays_d[3 * ray], d1 = rays_d[3 * ray + 1], d2 = rays_d[3 * ray + 2];
    const float q0 = grad[3 * (size_t)i], q1 = grad[3 * (size_t)i + 1], q2 = grad[3 * (size_t)i + 2];
    const float tc = d0 * q0 + d1 * q1 + d2 * q2;
    const float ic = -fmaxf(-tc, 0.f);
    const float sv = sdf[i];
    const float dist = (k + 1 < spr) ? z[i + 1] - z[i] : sample_dist;
    const float half = ic * dist * 0.5f;
    const float c = sigmoid_e((sv - half) * inv_s);
    const float nx = sigmoid_e((sv + half) * inv_s);
    const float A = (c - nx) + 1e-5f, B = c + 1e-5f;
    const float a_raw = A / B;
    const float ga = (a_raw > 0.f && a_raw < 1.f) ? g_alpha[i] : 0.f;   // clip(0, 1)
    const float gc = ga * (B - A) / (B * B);
    const float gnx = -ga / B;
    const float gx1 = gc * inv_s * c * (1.f - c);
    const float gx2 = gnx * inv_s * nx * (1.f - nx);
    float o_s = gx1 + gx2;
    const float ghalf = gx2 - gx1;
    const float gtc = (tc < 0.f) ? ghalf * dist * 0.5f : 0.f;
    float o_g[3] = {gtc * d0, gtc * d1, gtc * d2};
    // (the same association as the two launches this replaces: (alpha part) + direct, then + eikonal)
    if (g_sdf_out != nullptr) o_s += g_sdf_out[i];
    float a3[3] = {0.f, 0.f, 0.f};
    if (g_grad_out != nullptr) {
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) a3[cc] = g_grad_out[3 * (size_t)i + cc];
    }
    if (g_eik != nullptr) {
        const float nrm = sqrtf(q0 * q0 + q1 * q1 + q2 * q2);
        const float kk = (2.f / (float)n) * g_eik[0] * (nrm - 1.f) / fmaxf(nrm, 1e-30f);
        a3[0] += kk * q0;
        a3[1] += kk * q1;
        a3[2] += kk * q2;
    }
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) o_g[cc] += a3[cc];
    gs[i] = o_s;
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) gg[3 * (size_t)i + cc] = o_g[cc];
    if (pos != nullptr) {
        const int slot = pos[i];
        if (slot >= 0) {
            gs_c[slot] = o_s;
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
                gg_c[3 * (size_t)slot + cc] = o_g[cc];
                gr_c[3 * (size_t)slot + cc] = g_rgb[3 * (size_t)i + cc];
            }
        }
    }
    if (g_rays_d_s != nullptr) {   // per SAMPLE, summed per ray in sample order by the caller's next kernel (k_obj_rays_bwd): no atomics, the same bits in every run
        g_rays_d_s[3 * (size_t)i] = gtc * q0;
        g_rays_d_s[3 * (size_t)i + 1] = gtc * q1;
        g_rays_d_s[3 * (size_t)i + 2] = gtc * q2;
    } else if (g_rays_d != nullptr) {
        const int ray0 = __shfl(ray, 0, 64);
        const bool whole = __all(ray == ray0) && __popcll(__ballot(1)) == 64;
        if (whole) {
            const float r0 = wave_sum(gtc * q0), r1 = wave_sum(gtc * q1), r2 = wave_sum(gtc * q2);
            if ((threadIdx.x & 63) == 0) {
                atomicAdd(g_rays_d + 3 * ray, r0);
                atomicAdd(g_rays_d + 3 * ray + 1, r1);
                atomicAdd(g_rays_d + 3 * ray + 2, r2);
            }
        } else {
            atomicAdd(g_rays_d + 3 * ray, gtc * q0);
            atomicAdd(g_rays_d + 3 * ray + 1, gtc * q1);
            atomicAdd(g_rays_d + 3 * ray + 2, gtc * q2);
        }
    }
}
// g_rays_d (may be NULL) must be zeroed by the caller -- or listed in `zero_bufs`, when no sample's atomics can precede its zero
// fill: that holds only for buffers this launch does not accumulate into, so g_rays_d is NOT to be listed there.
int alpha_bwd_up(const float* sdf, const float* grad, const float* rays_d, const float* z, const float* g_alpha, int n, int spr, float sample_dist,
                 float inv_s, const float* g_sdf_out, const float* g_grad_out, const float* g_eik, float* gs, float* gg, float* g_rays_d,
                 const int* pos, const int* n_dev, const float* g_rgb, float* gs_c, float* gg_c, float* gr_c, float* const* zero_bufs,
                 const size_t* zero_sizes, int n_zero, hipStream_t s, float* g_rays_d_samples, const int* seg) {
    HN_REQUIRE(spr > 0 && n_zero <= 4, "samples_per_ray must be positive, at most four buffers to zero");
    if (n == 0) return HN_OK;
    AlphaUpZero zl{};
    int most = n;
    for (int b = 0; b < n_zero; ++b) {
        zl.p[b] = zero_bufs[b];
        zl.n[b] = (int)zero_sizes[b];
        most = most > zl.n[b] ? most : zl.n[b];
    }
    hipLaunchKernelGGL(k_alpha_bwd_up, dim3((most + 255) / 256), dim3(256), 0, s, sdf, grad, rays_d, z, g_alpha, n, spr, sample_dist, inv_s, g_sdf_out,
                       g_grad_out, g_eik, gs, gg, g_rays_d, pos, n_dev, g_rgb, gs_c, gg_c, gr_c, zl, g_rays_d_samples, seg);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

// d loss / d inv_s of the alpha stage (utils/renderer.py:144-161: inv_s = exp(10 variance) is a trained parameter,
// exp_runner.py:208-242): sum over the samples of gc c (1 - c) x1 / inv_s ... written on the pre-sigmoid arguments
// x1 = (s - half) inv_s, x2 = (s + half) inv_s.  One atomic per block (caller zeroes g_inv_s).
__global__ void k_alpha_inv_s_bwd(const float* __restrict__ sdf, const float* __restrict__ grad, const float* __restrict__ rays_d,
                                  const float* __restrict__ dists, const float* __restrict__ g_alpha, const float* __restrict__ g_c,
                                  int n, int spr, float inv_s_host, float* __restrict__ g_inv_s, const float* __restrict__ inv_s_dev) {
    __shared__ float part[4];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float inv_s = inv_s_dev != nullptr ? inv_s_dev[0] : inv_s_host;
    float v = 0.f;
    if (i < n) {
        const int ray = i / spr;
        const float tc = rays_d[3 * ray] * grad[3 * (size_t)i] + rays_d[3 * ray + 1] * grad[3 * (size_t)i + 1] +
                         rays_d[3 * ray + 2] * grad[3 * (size_t)i + 2];
        const float ic = -fmaxf(-tc, 0.f);
        const float s = sdf[i];
        const float half = ic * dists[i] * 0.5f;
        const float c = sigmoid_e((s - half) * inv_s);
        const float nx = sigmoid_e((s + half) * inv_s);
        const float A = (c - nx) + 1e-5f, B = c + 1e-5f;
        const float a_raw = A / B;
        const float ga = (a_raw > 0.f && a_raw < 1.f) ? g_alpha[i] : 0.f;
        const float gc = ga * (B - A) / (B * B) + (g_c != nullptr ? g_c[i] : 0.f);
        const float gnx = -ga / B;
        v = gc * c * (1.f - c) * (s - half) + gnx * nx * (1.f - nx) * (s + half);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(g_inv_s, part[0] + part[1] + part[2] + part[3]);
}
int alpha_inv_s_bwd(const float* sdf, const float* grad, const float* rays_d, const float* dists, const float* g_alpha,
                    const float* g_c, int n, int spr, float inv_s, float* g_inv_s, hipStream_t s, const float* inv_s_dev) {
    HN_REQUIRE(spr > 0 && g_inv_s != nullptr, "bad arguments");
    if (n == 0) return HN_OK;
    hipLaunchKernelGGL(k_alpha_inv_s_bwd, dim3((n + 255) / 256), dim3(256), 0, s, sdf, grad, rays_d, dists, g_alpha, g_c, n, spr,
                       inv_s, g_inv_s, inv_s_dev);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

int composite1_bwd(const float* alpha_in, const float* c, const float* rgb, const float* g_color, const float* g_wsum,
                   int n_rays, int S, float* g_alpha, float* g_c, float* g_rgb, hipStream_t s) {
    HN_REQUIRE(S >= 1, "S must be positive");
    if (n_rays == 0) return HN_OK;
    const dim3 wgrid((n_rays + 3) / 4);   // 4 rays (waves) per block
    if (S <= 64)
        hipLaunchKernelGGL(k_composite1_bwd_wave<1>, wgrid, dim3(256), 0, s, alpha_in, c, rgb, g_color, g_wsum, n_rays, S, g_alpha, g_c, g_rgb);
    else if (S <= 128)
        hipLaunchKernelGGL(k_composite1_bwd_wave<2>, wgrid, dim3(256), 0, s, alpha_in, c, rgb, g_color, g_wsum, n_rays, S, g_alpha, g_c, g_rgb);
    else if (S <= 192)
        hipLaunchKernelGGL(k_composite1_bwd_wave<3>, wgrid, dim3(256), 0, s, alpha_in, c, rgb, g_color, g_wsum, n_rays, S, g_alpha, g_c, g_rgb);
    else if (S <= 256)
        hipLaunchKernelGGL(k_composite1_bwd_wave<4>, wgrid, dim3(256), 0, s, alpha_in, c, rgb, g_color, g_wsum, n_rays, S, g_alpha, g_c, g_rgb);
    else
        hipLaunchKernelGGL(k_composite1_bwd, dim3((n_rays + 63) / 64), dim3(64), 0, s, alpha_in, c, rgb, g_color, g_wsum, n_rays, S, g_alpha, g_c,
                           g_rgb);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

int composite2_bwd(const float* ah, const float* rgbh, const float* ao, const float* rgbo, const float* g_color,
                   const float* g_wsum, int n_rays, int S, float* g_ah, float* g_rgbh, float* g_ao, float* g_rgbo,
                   hipStream_t s) {
    HN_REQUIRE(S >= 1, "S must be positive");
    if (n_rays == 0) return HN_OK;
    const dim3 wgrid((n_rays + 3) / 4);   // 4 rays (waves) per block
    if (S <= 64)
        hipLaunchKernelGGL(k_composite2_bwd_wave<1>, wgrid, dim3(256), 0, s, ah, rgbh, ao, rgbo, g_color, g_wsum, n_rays, S, g_ah, g_rgbh, g_ao, g_rgbo);
    else if (S <= 128)
        hipLaunchKernelGGL(k_composite2_bwd_wave<2>, wgrid, dim3(256), 0, s, ah, rgbh, ao, rgbo, g_color, g_wsum, n_rays, S, g_ah, g_rgbh, g_ao, g_rgbo);
    else if (S <= 192)
        hipLaunchKernelGGL(k_composite2_bwd_wave<3>, wgrid, dim3(256), 0, s, ah, rgbh, ao, rgbo, g_color, g_wsum, n_rays, S, g_ah, g_rgbh, g_ao, g_rgbo);
    else if (S <= 256)
        hipLaunchKernelGGL(k_composite2_bwd_wave<4>, wgrid, dim3(256), 0, s, ah, rgbh, ao, rgbo, g_color, g_wsum, n_rays, S, g_ah, g_rgbh, g_ao, g_rgbo);
    else
        hipLaunchKernelGGL(k_composite2_bwd, dim3((n_rays + 63) / 64), dim3(64), 0, s, ah, rgbh, ao, rgbo, g_color, g_wsum,
                           n_rays, S, g_ah, g_rgbh, g_ao, g_rgbo);
    HN_LAUNCH_CHECK();
    return HN_OK;
}


// ---- render-dependent loss terms of a fitting step (fitting_single.py:251-283; fitting_video.py:285-309) ------------
// colour: sum |(color - true_rgb) mask| / R;  mask: mean BCE(clip(weight_sum, 1e-3, 1 - 1e-3), mask);  contact: mean of
// |s_h| + |s_o| where that sum < 1e-2;  penetration: mean of the same over s_o < 0 and s_h < 0.  One launch for the
// sums (sums[6] = colour loss, bce loss (both already / R), contact sum, contact count, penetration sum, penetration
// count; zeroed here) and one for
// all four gradients, instead of ~35 + ~50 element-wise launches in a step that is a chain of dependent launches.
__global__ void k_fit_loss_sums(const float* __restrict__ color, const float* __restrict__ wsum, const float* __restrict__ true_rgb,
                                const float* __restrict__ true_mask, int n_rays, const float* __restrict__ sdf_h,
                                const float* __restrict__ sdf_o, int n_samples, float* __restrict__ sums) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (i < n_rays) {
        const float m = true_mask[i], inv = 1.f / (float)n_rays;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[0] += fabsf((color[3 * (size_t)i + c] - true_rgb[3 * (size_t)i + c]) * m);
        const float w = fminf(fmaxf(wsum[i], 1e-3f), 1.f - 1e-3f);
        v[0] *= inv;
        v[1] = -(m * logf(w) + (1.f - m) * logf(1.f - w)) * inv;
    }
    if (sdf_h != nullptr && i < n_samples) {
        const float sh = sdf_h[i], so = sdf_o[i];
        const float a = fabsf(sh) + fabsf(so);
        if (a < 1e-2f) {
            v[2] = a;
            v[3] = 1.f;
        }
        if (so < 0.f && sh < 0.f) {
            v[4] = a;
            v[5] = 1.f;
        }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const float t = wave_sum(v[k]);
        if ((threadIdx.x & 63) == 0 && t != 0.f) atomicAdd(sums + k, t);
    }
}
// g[4] = upstream gradients of (colour, mask, contact, penetration) losses (device scalars)
__global__ void k_fit_loss_grads(const float* __restrict__ color, const float* __restrict__ wsum, const float* __restrict__ true_rgb,
                                 const float* __restrict__ true_mask, int n_rays, const float* __restrict__ sdf_h,
                                 const float* __restrict__ sdf_o, int n_samples, const float* __restrict__ sums,
                                 const float* __restrict__ g, float* __restrict__ g_color, float* __restrict__ g_wsum,
                                 float* __restrict__ g_sdf_h, float* __restrict__ g_sdf_o) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_rays) {
        const float m = true_mask[i], inv = 1.f / (float)n_rays;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float e = (color[3 * (size_t)i + c] - true_rgb[3 * (size_t)i + c]) * m;
            g_color[3 * (size_t)i + c] = g[0] * inv * m * (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f));
        }
        const float w0 = wsum[i];
        const float w = fminf(fmaxf(w0, 1e-3f), 1.f - 1e-3f);
        const bool inside = w0 >= 1e-3f && w0 <= 1.f - 1e-3f;   // clip passes the gradient on its closed range, as torch
        g_wsum[i] = inside ? g[1] * inv * (w - m) / (w * (1.f - w)) : 0.f;
    }
    if (sdf_h != nullptr && i < n_samples) {
        const float sh = sdf_h[i], so = sdf_o[i];
        const float a = fabsf(sh) + fabsf(so);
        float k = 0.f;
        if (a < 1e-2f) k += g[2] / (sums[3] + 1e-9f);
        if (so < 0.f && sh < 0.f) k += g[3] / (sums[5] + 1e-9f);
        g_sdf_h[i] = k * (sh > 0.f ? 1.f : (sh < 0.f ? -1.f : 0.f));
        g_sdf_o[i] = k * (so > 0.f ? 1.f : (so < 0.f ? -1.f : 0.f));
    }
}
int fit_loss_sums(const float* color, const float* wsum, const float* true_rgb, const float* true_mask, int n_rays, const float* sdf_h,
                  const float* sdf_o, int n_samples, float* sums6, hipStream_t s) {
    HN_REQUIRE(color && wsum && true_rgb && true_mask && sums6 && n_rays >= 0 && (sdf_h == nullptr) == (sdf_o == nullptr), "bad arguments");
    HN_CHECK_HIP(hipMemsetAsync(sums6, 0, 6 * sizeof(float), s));
    const int n = n_rays > n_samples ? n_rays : (sdf_h != nullptr ? n_samples : n_rays);
    if (n <= 0) return HN_OK;
    hipLaunchKernelGGL(k_fit_loss_sums, dim3((n + 255) / 256), dim3(256), 0, s, color, wsum, true_rgb, true_mask, n_rays, sdf_h, sdf_o,
                       sdf_h != nullptr ? n_samples : 0, sums6);
    HN_LAUNCH_CHECK();
    return HN_OK;
}
int fit_loss_grads(const float* color, const float* wsum, const float* true_rgb, const float* true_mask, int n_rays, const float* sdf_h,
                   const float* sdf_o, int n_samples, const float* sums6, const float* g4, float* g_color, float* g_wsum,
                   float* g_sdf_h, float* g_sdf_o, hipStream_t s) {
    HN_REQUIRE(color && wsum && true_rgb && true_mask && sums6 && g4 && g_color && g_wsum, "bad arguments");
    HN_REQUIRE(sdf_h == nullptr || (sdf_o && g_sdf_h && g_sdf_o), "sdf gradients need both fields");
    const int n = n_rays > n_samples ? n_rays : (sdf_h != nullptr ? n_samples : n_rays);
    if (n <= 0) return HN_OK;
    hipLaunchKernelGGL(k_fit_loss_grads, dim3((n + 255) / 256), dim3(256), 0, s, color, wsum, true_rgb, true_mask, n_rays, sdf_h, sdf_o,
                       sdf_h != nullptr ? n_samples : 0, sums6, g4, g_color, g_wsum, g_sdf_h, g_sdf_o);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

// ---- the whole loss of a fitting_single step in one launch (fitting_single.py:251-288) ------------------------------------
// terms8 = {loss, colour, mask, contact, penetration, joint, verts, 0} from the sums of k_fit_loss_sums, the vertex loss of
// k_verts_loss and the joint loss  sum_j |joint3d_pred_j - joint_3d_j| / n_joints  (pose_loss, :119-122), combined as
//   loss = w0 (colour + 0.5 mask) + w1 contact + w2 penetration + w3 joint + w4 verts
// (w = {1, 30, 20, 30, 20} for fit type 12, {1, 0, 0, 100, 5} for fit type 1).  g_joint [n_joints,3] = d joint / d joint_3d.
// As torch operators this was ~25 element-wise launches forward and ~35 backward in a step that is a chain of dependent launches.
__global__ __launch_bounds__(64) void k_fit_total(const float* __restrict__ sums6, const float* __restrict__ verts_loss,
                                                  const float* __restrict__ joint_3d, const float* __restrict__ joint_pred, int n_joints,
                                                  float w0, float w1, float w2, float w3, float w4, float* __restrict__ terms8,
                                                  float* __restrict__ g_joint) {
    const int j = threadIdx.x;
    float nrm = 0.f;
    if (j < n_joints) {
        const float e0 = joint_3d[3 * j] - joint_pred[3 * j], e1 = joint_3d[3 * j + 1] - joint_pred[3 * j + 1],
                    e2 = joint_3d[3 * j + 2] - joint_pred[3 * j + 2];
        nrm = sqrtf(e0 * e0 + e1 * e1 + e2 * e2);
        const float inv = nrm > 0.f ? 1.f / (nrm * (float)n_joints) : 0.f;   // torch.norm's subgradient at 0 is 0
        g_joint[3 * j] = e0 * inv;
        g_joint[3 * j + 1] = e1 * inv;
        g_joint[3 * j + 2] = e2 * inv;
    }
    const float joint = wave_sum(nrm) / (float)n_joints;
    if (j == 0) {
        const float colour = sums6[0], mask = sums6[1];
        const float contact = sums6[2] / (sums6[3] + 1e-9f), penet = sums6[4] / (sums6[5] + 1e-9f);
        const float verts = verts_loss[0];
        terms8[0] = w0 * (colour + 0.5f * mask) + (w1 * contact + w2 * penet) + (w3 * joint + w4 * verts);
        terms8[1] = colour;
        terms8[2] = mask;
        terms8[3] = contact;
        terms8[4] = penet;
        terms8[5] = joint;
        terms8[6] = verts;
        terms8[7] = 0.f;
    }
}
// upstream gradient of the loss (a device scalar) -> g4 for k_fit_loss_grads and the scaled pose-side gradients
__global__ __launch_bounds__(64) void k_fit_total_bwd(const float* __restrict__ g_loss, float w0, float w1, float w2, float w3, float w4,
                                                      const float* __restrict__ g_joint, const float* __restrict__ gR,
                                                      const float* __restrict__ gt, int n_joints, float* __restrict__ g4,
                                                      float* __restrict__ g_joint_out, float* __restrict__ gR_out, float* __restrict__ gt_out) {
    const float g = g_loss[0];
    const int i = threadIdx.x;
    if (i == 0) {
        g4[0] = g * w0;
        g4[1] = g * w0 * 0.5f;
        g4[2] = g * w1;
        g4[3] = g * w2;
    }
    for (int k = i; k < 3 * n_joints; k += 64) g_joint_out[k] = g * w3 * g_joint[k];
    if (i < 9) gR_out[i] = g * w4 * gR[i];
    if (i < 3) gt_out[i] = g * w4 * gt[i];
}
int fit_total(const float* sums6, const float* verts_loss, const float* joint_3d, const float* joint_pred, int n_joints, const float* w5,
              float* terms8, float* g_joint, hipStream_t s) {
    HN_REQUIRE(sums6 && verts_loss && joint_3d && joint_pred && w5 && terms8 && g_joint && n_joints >= 1 && n_joints <= 64, "bad arguments");
    hipLaunchKernelGGL(k_fit_total, dim3(1), dim3(64), 0, s, sums6, verts_loss, joint_3d, joint_pred, n_joints, w5[0], w5[1], w5[2], w5[3], w5[4],
                       terms8, g_joint);
    HN_LAUNCH_CHECK();
    return HN_OK;
}
int fit_total_bwd(const float* g_loss, const float* w5, const float* g_joint, const float* gR, const float* gt, int n_joints, float* g4,
                  float* g_joint_out, float* gR_out, float* gt_out, hipStream_t s) {
    HN_REQUIRE(g_loss && w5 && g_joint && gR && gt && g4 && g_joint_out && gR_out && gt_out && n_joints >= 1 && n_joints <= 64, "bad arguments");
    hipLaunchKernelGGL(k_fit_total_bwd, dim3(1), dim3(64), 0, s, g_loss, w5[0], w5[1], w5[2], w5[3], w5[4], g_joint, gR, gt, n_joints, g4,
                       g_joint_out, gR_out, gt_out);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

constexpr int FIT_LOSS_MAX_FRAMES = 16;
// ---- the whole loss of a fitting_single step as ONE launch forward and ONE backward (fitting_single.py:251-288) -------------
// Forward (k_fit_step_loss): every block reduces its share of the six sums of k_fit_loss_sums to a slot of `partials`; the block
// that finishes last (a counter it resets for the next call) adds the slots IN INDEX ORDER -- the sums do not depend on the order
// the blocks ran in: two runs of a step give the same loss terms bit for bit -- and goes on to the vertex loss (k_verts_loss's
// statements for one pose pair), the joint loss and the weighted total (k_fit_total's).  As separate launches (memset, sums,
// vertex loss, total) it was four links of the dependent-launch chain between the render and its backward pass.
// Several frames in one launch (blockIdx.y = frame): every frame is its own problem -- its own blocks, slots, counter and last block,
// the arithmetic and the order of every sum those of a one-frame launch on that frame's planes (frame f of every per-frame array: base +
// f x its width; the object's vertices per frame through FrameVerts).
struct FrameVerts {
    const float* p[FIT_LOSS_MAX_FRAMES];
    int n[FIT_LOSS_MAX_FRAMES];
};
__global__ __launch_bounds__(256) void k_fit_step_loss(const float* __restrict__ color, const float* __restrict__ wsum, const float* __restrict__ true_rgb,
                                                       const float* __restrict__ true_mask, int n_rays, const float* __restrict__ sdf_h,
                                                       const float* __restrict__ sdf_o, int n_samples, const float* __restrict__ joint_3d,
                                                       const float* __restrict__ joint_pred, int n_joints, const float* __restrict__ Ra,
                                                       const float* __restrict__ ta, const float* __restrict__ Rb, const float* __restrict__ tb,
                                                       const FrameVerts fv, float w0, float w1, float w2, float w3, float w4,
                                                       float* __restrict__ scratch, int scratch_floats,
                                                       float* __restrict__ sums6, float* __restrict__ terms8, float* __restrict__ g_joint,
                                                       float* __restrict__ gR, float* __restrict__ gt) {
    __shared__ float red[13][4];
    __shared__ bool is_last;
    const int fr = blockIdx.y;
    color += 3 * (size_t)fr * n_rays, true_rgb += 3 * (size_t)fr * n_rays, wsum += (size_t)fr * n_rays, true_mask += (size_t)fr * n_rays;
    if (sdf_h != nullptr) sdf_h += (size_t)fr * n_samples, sdf_o += (size_t)fr * n_samples;
    joint_3d += 3 * fr * n_joints, joint_pred += 3 * fr * n_joints, g_joint += 3 * fr * n_joints;
    Ra += 9 * fr, Rb += 9 * fr, ta += 3 * fr, tb += 3 * fr, gR += 9 * fr, gt += 3 * fr, sums6 += 6 * fr, terms8 += 8 * fr;
    const float* __restrict__ verts = fv.p[fr];
    const int n_verts = fv.n[fr];
    scratch += (size_t)fr * scratch_floats;
    float* __restrict__ partials = scratch + 16;
    unsigned* __restrict__ counter = reinterpret_cast<unsigned*>(scratch);
    float* __restrict__ pose2 = scratch + 4;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (i < n_rays) {
        const float m = true_mask[i], inv = 1.f / (float)n_rays;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[0] += fabsf((color[3 * (size_t)i + c] - true_rgb[3 * (size_t)i + c]) * m);
        const float w = fminf(fmaxf(wsum[i], 1e-3f), 1.f - 1e-3f);
        v[0] *= inv;
        v[1] = -(m * logf(w) + (1.f - m) * logf(1.f - w)) * inv;
    }
    if (sdf_h != nullptr && i < n_samples) {
        const float sh = sdf_h[i], so = sdf_o[i];
        const float a = fabsf(sh) + fabsf(so);
        if (a < 1e-2f) {
            v[2] = a;
            v[3] = 1.f;
        }
        if (so < 0.f && sh < 0.f) {
            v[4] = a;
            v[5] = 1.f;
        }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const float t = wave_sum(v[k]);
        if (lane == 0) red[k][wave] = t;
    }
    __syncthreads();
    if (threadIdx.x < 6) partials[6 * (size_t)blockIdx.x + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
    __syncthreads();
    if (blockIdx.x == 0) {
        // ---- the pose part (independent of the render: block 0 does it while the other blocks reduce their sums): the vertex loss
        //      of the pose pair (k_verts_loss, one pair) with its gradient w.r.t. (Ra, ta), and the joint loss (k_fit_total's statements)
        float D[9], dd[3];
#pragma unroll
        for (int k = 0; k < 9; ++k) D[k] = Ra[k] - Rb[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) dd[k] = ta[k] - tb[k];
        float a13[13];
#pragma unroll
        for (int k = 0; k < 13; ++k) a13[k] = 0.f;
        for (int q = threadIdx.x; q < n_verts; q += blockDim.x) {
            const float x = verts[3 * q], y = verts[3 * q + 1], z = verts[3 * q + 2];
            const float e0 = D[0] * x + D[1] * y + D[2] * z + dd[0], e1 = D[3] * x + D[4] * y + D[5] * z + dd[1], e2 = D[6] * x + D[7] * y + D[8] * z + dd[2];
            const float nn = sqrtf(e0 * e0 + e1 * e1 + e2 * e2);
            const float inv = nn > 0.f ? 1.f / nn : 0.f;
            const float u0 = e0 * inv, u1 = e1 * inv, u2 = e2 * inv;
            a13[0] += nn;
            a13[1] += u0 * x; a13[2] += u0 * y; a13[3] += u0 * z;
            a13[4] += u1 * x; a13[5] += u1 * y; a13[6] += u1 * z;
            a13[7] += u2 * x; a13[8] += u2 * y; a13[9] += u2 * z;
            a13[10] += u0; a13[11] += u1; a13[12] += u2;
        }
#pragma unroll
        for (int k = 0; k < 13; ++k) {
            const float t = wave_sum(a13[k]);
            if (lane == 0) red[k][wave] = t;
        }
        __syncthreads();
        if (threadIdx.x < 13) {
            const float t = (red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]) / (float)n_verts;
            if (threadIdx.x == 0)
                pose2[1] = t;
            else if (threadIdx.x < 10)
                gR[threadIdx.x - 1] = t;
            else
                gt[threadIdx.x - 10] = t;
        }
        if (wave == 0) {
            float nrm = 0.f;
            if (lane < n_joints) {
                const float e0 = joint_3d[3 * lane] - joint_pred[3 * lane], e1 = joint_3d[3 * lane + 1] - joint_pred[3 * lane + 1],
                            e2 = joint_3d[3 * lane + 2] - joint_pred[3 * lane + 2];
                nrm = sqrtf(e0 * e0 + e1 * e1 + e2 * e2);
                const float inv = nrm > 0.f ? 1.f / (nrm * (float)n_joints) : 0.f;
                g_joint[3 * lane] = e0 * inv;
                g_joint[3 * lane + 1] = e1 * inv;
                g_joint[3 * lane + 2] = e2 * inv;
            }
            const float joint = wave_sum(nrm) / (float)n_joints;
            if (lane == 0) pose2[0] = joint;
        }
        __syncthreads();
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) is_last = atomicAdd(counter, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    // ---- the last block: the six sums over the slots -- thread t the slots t, t + 256, .., then the fixed tree of wave_sum and the
    //      four waves in order: the result does not depend on which block came last or in what order the others ran
    {
        float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const volatile float* ps = reinterpret_cast<const volatile float*>(partials);
        for (unsigned b = threadIdx.x; b < gridDim.x; b += 256) {
#pragma unroll
            for (int k = 0; k < 6; ++k) acc[k] += ps[6 * (size_t)b + k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float t = wave_sum(acc[k]);
            if (lane == 0) red[k][wave] = t;
        }
        __syncthreads();
        if (threadIdx.x < 6) {
            const float t = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
            sums6[threadIdx.x] = t;
            red[threadIdx.x][0] = t;
        }
        if (threadIdx.x == 0) *counter = 0u;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float colour = red[0][0], mask = red[1][0];
        const float contact = red[2][0] / (red[3][0] + 1e-9f), penet = red[4][0] / (red[5][0] + 1e-9f);
        const volatile float* pz = reinterpret_cast<const volatile float*>(pose2);
        const float joint = pz[0], verts_loss = pz[1];
        terms8[0] = w0 * (colour + 0.5f * mask) + (w1 * contact + w2 * penet) + (w3 * joint + w4 * verts_loss);
        terms8[1] = colour;
        terms8[2] = mask;
        terms8[3] = contact;
        terms8[4] = penet;
        terms8[5] = joint;
        terms8[6] = verts_loss;
        terms8[7] = 0.f;
    }
}
// Backward: k_fit_loss_grads with the four upstream factors formed from the loss's upstream gradient and the weights in the
// kernel (k_fit_total_bwd's statements), and the pose-side gradients scaled by block 0.
__global__ __launch_bounds__(256) void k_fit_step_loss_bwd(const float* __restrict__ color, const float* __restrict__ wsum, const float* __restrict__ true_rgb,
                                                           const float* __restrict__ true_mask, int n_rays, const float* __restrict__ sdf_h,
                                                           const float* __restrict__ sdf_o, int n_samples, const float* __restrict__ sums,
                                                           const float* __restrict__ g_loss, float w0, float w1, float w2, float w3, float w4,
                                                           const float* __restrict__ g_joint, const float* __restrict__ gR, const float* __restrict__ gt,
                                                           int n_joints, float* __restrict__ g_color, float* __restrict__ g_wsum,
                                                           float* __restrict__ g_sdf_h, float* __restrict__ g_sdf_o, float* __restrict__ g_joint_out,
                                                           float* __restrict__ gR_out, float* __restrict__ gt_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int fr = blockIdx.y;      // (frame f of every per-frame array: base + f x its width; g_loss is the one upstream scalar)
    color += 3 * (size_t)fr * n_rays, true_rgb += 3 * (size_t)fr * n_rays, wsum += (size_t)fr * n_rays, true_mask += (size_t)fr * n_rays;
    g_color += 3 * (size_t)fr * n_rays, g_wsum += (size_t)fr * n_rays;
    if (sdf_h != nullptr) sdf_h += (size_t)fr * n_samples, sdf_o += (size_t)fr * n_samples, g_sdf_h += (size_t)fr * n_samples, g_sdf_o += (size_t)fr * n_samples;
    sums += 6 * fr, g_joint += 3 * fr * n_joints, g_joint_out += 3 * fr * n_joints, gR += 9 * fr, gt += 3 * fr, gR_out += 9 * fr, gt_out += 3 * fr;
    const float gl = g_loss[0];
    const float g0 = gl * w0, g1 = gl * w0 * 0.5f, g2 = gl * w1, g3 = gl * w2;
    if (blockIdx.x == 0) {
        for (int k = threadIdx.x; k < 3 * n_joints; k += blockDim.x) g_joint_out[k] = gl * w3 * g_joint[k];
        if (threadIdx.x < 9) gR_out[threadIdx.x] = gl * w4 * gR[threadIdx.x];
        if (threadIdx.x < 3) gt_out[threadIdx.x] = gl * w4 * gt[threadIdx.x];
    }
    if (i < n_rays) {
        const float m = true_mask[i], inv = 1.f / (float)n_rays;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float e = (color[3 * (size_t)i + c] - true_rgb[3 * (size_t)i + c]) * m;
            g_color[3 * (size_t)i + c] = g0 * inv * m * (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f));
        }
        const float wv = wsum[i];
        const float w = fminf(fmaxf(wv, 1e-3f), 1.f - 1e-3f);
        const bool inside = wv >= 1e-3f && wv <= 1.f - 1e-3f;
        g_wsum[i] = inside ? g1 * inv * (w - m) / (w * (1.f - w)) : 0.f;
    }
    if (sdf_h != nullptr && i < n_samples) {
        const float sh = sdf_h[i], so = sdf_o[i];
        const float a = fabsf(sh) + fabsf(so);
        float k = 0.f;
        if (a < 1e-2f) k += g2 / (sums[3] + 1e-9f);
        if (so < 0.f && sh < 0.f) k += g3 / (sums[5] + 1e-9f);
        g_sdf_h[i] = k * (sh > 0.f ? 1.f : (sh < 0.f ? -1.f : 0.f));
        g_sdf_o[i] = k * (so > 0.f ? 1.f : (so < 0.f ? -1.f : 0.f));
    }
}
size_t fit_step_loss_scratch_bytes(int n_rays, int n_samples) {
    const int n = n_rays > n_samples ? n_rays : n_samples;
    return ((size_t)((n + 255) / 256 + 1) * 6 + 16) * sizeof(float);
}
int fit_step_loss_frames(int n_frames, const float* color, const float* wsum, const float* true_rgb, const float* true_mask, int n_rays, const float* sdf_h,
                         const float* sdf_o, int n_samples, const float* joint_3d, const float* joint_pred, int n_joints, const float* Ra, const float* ta,
                         const float* Rb, const float* tb, const float* const* verts, const int* n_verts, const float* w5, void* scratch, size_t scratch_bytes,
                         float* sums6, float* terms8, float* g_joint, float* gR, float* gt, hipStream_t s) {
    HN_REQUIRE(color && wsum && true_rgb && true_mask && joint_3d && joint_pred && Ra && ta && Rb && tb && verts && n_verts && w5 && scratch && sums6 &&
                   terms8 && g_joint && gR && gt,
               "fit_step_loss: null argument");
    HN_REQUIRE(n_frames >= 1 && n_frames <= FIT_LOSS_MAX_FRAMES, "fit_step_loss: 1 .. %d frames per launch", FIT_LOSS_MAX_FRAMES);
    HN_REQUIRE(n_rays >= 1 && (sdf_h == nullptr) == (sdf_o == nullptr) && n_joints >= 1 && n_joints <= 64, "fit_step_loss: bad sizes");
    const int ns = sdf_h != nullptr ? n_samples : 0;
    const size_t per_frame = fit_step_loss_scratch_bytes(n_rays, ns);
    HN_REQUIRE(scratch_bytes >= per_frame * (size_t)n_frames, "fit_step_loss: scratch too small");
    FrameVerts fv{};
    for (int f = 0; f < n_frames; ++f) {
        HN_REQUIRE(verts[f] != nullptr && n_verts[f] >= 1, "fit_step_loss: the vertices of frame %d", f);
        fv.p[f] = verts[f];
        fv.n[f] = n_verts[f];
    }
    const int n = n_rays > ns ? n_rays : ns;
    const int blocks = (n + 255) / 256;
    // (per frame: counter at float 0 -- zero when the scratch is first handed over, every launch leaves it zero --, pose part at 4, slots at 16)
    hipLaunchKernelGGL(k_fit_step_loss, dim3(blocks, n_frames), dim3(256), 0, s, color, wsum, true_rgb, true_mask, n_rays, sdf_h, sdf_o, ns, joint_3d,
                       joint_pred, n_joints, Ra, ta, Rb, tb, fv, w5[0], w5[1], w5[2], w5[3], w5[4], reinterpret_cast<float*>(scratch),
                       (int)(per_frame / sizeof(float)), sums6, terms8, g_joint, gR, gt);
    HN_LAUNCH_CHECK();
    return HN_OK;
}
int fit_step_loss(const float* color, const float* wsum, const float* true_rgb, const float* true_mask, int n_rays, const float* sdf_h, const float* sdf_o,
                  int n_samples, const float* joint_3d, const float* joint_pred, int n_joints, const float* Ra, const float* ta, const float* Rb,
                  const float* tb, const float* verts, int n_verts, const float* w5, void* scratch, size_t scratch_bytes, float* sums6, float* terms8,
                  float* g_joint, float* gR, float* gt, hipStream_t s) {
    return fit_step_loss_frames(1, color, wsum, true_rgb, true_mask, n_rays, sdf_h, sdf_o, n_samples, joint_3d, joint_pred, n_joints, Ra, ta, Rb, tb, &verts,
                                &n_verts, w5, scratch, scratch_bytes, sums6, terms8, g_joint, gR, gt, s);
}
int fit_step_loss_bwd_frames(int n_frames, const float* color, const float* wsum, const float* true_rgb, const float* true_mask, int n_rays,
                             const float* sdf_h, const float* sdf_o, int n_samples, const float* sums6, const float* g_loss, const float* w5,
                             const float* g_joint, const float* gR, const float* gt, int n_joints, float* g_color, float* g_wsum, float* g_sdf_h,
                             float* g_sdf_o, float* g_joint_out, float* gR_out, float* gt_out, hipStream_t s) {
    HN_REQUIRE(color && wsum && true_rgb && true_mask && sums6 && g_loss && w5 && g_joint && gR && gt && g_color && g_wsum && g_joint_out && gR_out && gt_out,
               "fit_step_loss_bwd: null argument");
    HN_REQUIRE(n_frames >= 1 && n_frames <= FIT_LOSS_MAX_FRAMES, "fit_step_loss_bwd: 1 .. %d frames per launch", FIT_LOSS_MAX_FRAMES);
    HN_REQUIRE(sdf_h == nullptr || (sdf_o && g_sdf_h && g_sdf_o), "sdf gradients need both fields");
    const int ns = sdf_h != nullptr ? n_samples : 0;
    const int n = n_rays > ns ? n_rays : ns;
    if (n <= 0) return HN_OK;
    hipLaunchKernelGGL(k_fit_step_loss_bwd, dim3((n + 255) / 256, n_frames), dim3(256), 0, s, color, wsum, true_rgb, true_mask, n_rays, sdf_h, sdf_o, ns,
                       sums6, g_loss, w5[0], w5[1], w5[2], w5[3], w5[4], g_joint, gR, gt, n_joints, g_color, g_wsum, g_sdf_h, g_sdf_o, g_joint_out, gR_out,
                       gt_out);
    HN_LAUNCH_CHECK();
    return HN_OK;
}
int fit_step_loss_bwd(const float* color, const float* wsum, const float* true_rgb, const float* true_mask, int n_rays, const float* sdf_h,
                      const float* sdf_o, int n_samples, const float* sums6, const float* g_loss, const float* w5, const float* g_joint, const float* gR,
                      const float* gt, int n_joints, float* g_color, float* g_wsum, float* g_sdf_h, float* g_sdf_o, float* g_joint_out, float* gR_out,
                      float* gt_out, hipStream_t s) {
    return fit_step_loss_bwd_frames(1, color, wsum, true_rgb, true_mask, n_rays, sdf_h, sdf_o, n_samples, sums6, g_loss, w5, g_joint, gR, gt, n_joints,
                                    g_color, g_wsum, g_sdf_h, g_sdf_o, g_joint_out, gR_out, gt_out, s);
}

// ---- the loss of a training iteration (exp_runner.py:202-212, without the VGG term) as ONE launch forward and ONE backward ------
//   m = (true_mask > 0.5);  mask_sum = sum m + 1e-5;  colour = sum |(c - t) m| / mask_sum;  mask = BCE(clip(w, 1e-3, 1 - 1e-3), m) (mean);
//   psnr = 20 log10(1 / sqrt(sum (c - t)^2 m / (3 mask_sum)));  loss = colour + mask_weight mask + igr_weight eikonal.
// As torch operators it is ~22 launches forward and ~25 backward of ~5 us each, between the render's final evaluation and its adjoint: 0.2 ms of a
// 4 ms iteration.  One block: the sums are formed in a fixed order (thread t the rays t, t + 1024, ..; the xor tree; the waves in order).
// terms6 = loss, colour, mask, eikonal, psnr, mask_sum.
__global__ __launch_bounds__(1024) void k_train_loss(const float* __restrict__ color, const float* __restrict__ wsum, const float* __restrict__ gerr,
                                                     const float* __restrict__ true_rgb, const float* __restrict__ true_mask, int n_rays, float igr_weight,
                                                     float mask_weight, float* __restrict__ terms6) {
    __shared__ float red[4][16];
    float v[4] = {0.f, 0.f, 0.f, 0.f};   // mask count, sum |e m|, sum e^2 m, BCE sum
    for (int i = threadIdx.x; i < n_rays; i += blockDim.x) {
        const float m = true_mask[i] > 0.5f ? 1.f : 0.f;
        v[0] += m;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float e = color[3 * (size_t)i + c] - true_rgb[3 * (size_t)i + c];
            v[1] += fabsf(e * m);
            v[2] += e * e * m;
        }
        const float w = fminf(fmaxf(wsum[i], 1e-3f), 1.f - 1e-3f);
        v[3] -= m * logf(w) + (1.f - m) * logf(1.f - w);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float t = wave_sum(v[k]);
        if (lane == 0) red[k][wave] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            t[k] = 0.f;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t[k] += red[k][w];
        }
        const float mask_sum = t[0] + 1e-5f;
        const float colour = t[1] / mask_sum, mask = t[3] / (float)n_rays, eik = gerr[0];
        terms6[0] = colour + mask * mask_weight + eik * igr_weight;
        terms6[1] = colour;
        terms6[2] = mask;
        terms6[3] = eik;
        terms6[4] = 20.f * log10f(1.f / sqrtf(t[2] / (mask_sum * 3.f)));
        terms6[5] = mask_sum;
    }
}
__global__ __launch_bounds__(256) void k_train_loss_bwd(const float* __restrict__ color, const float* __restrict__ wsum, const float* __restrict__ true_rgb,
                                                        const float* __restrict__ true_mask, int n_rays, const float* __restrict__ terms6,
                                                        const float* __restrict__ g_loss, float igr_weight, float mask_weight, float* __restrict__ g_color,
                                                        float* __restrict__ g_wsum, float* __restrict__ g_gerr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float gl = g_loss[0];
    if (i == 0) g_gerr[0] = gl * igr_weight;
    if (i >= n_rays) return;
    const float m = true_mask[i] > 0.5f ? 1.f : 0.f;
    const float gc = gl / terms6[5];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float e = (color[3 * (size_t)i + c] - true_rgb[3 * (size_t)i + c]) * m;
        g_color[3 * (size_t)i + c] = gc * m * (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f));
    }
    const float wv = wsum[i];
    const float w = fminf(fmaxf(wv, 1e-3f), 1.f - 1e-3f);
    const bool inside = wv >= 1e-3f && wv <= 1.f - 1e-3f;
    g_wsum[i] = inside ? gl * mask_weight * (w - m) / fmaxf((1.f - w) * w, 1e-12f) / (float)n_rays : 0.f;
}
// SingleVarianceNetwork (utils/fields.py:248-249, utils/renderer.py:144): inv_s = clip(exp(10 variance), 1e-6, 1e6) and its chain rule
// g_variance = g_inv_s x (10 inv_s inside the clip range, else 0) on device scalars: one tiny launch each where torch operators take 3 / 7
// (a training iteration re-forms inv_s after every optimiser step and needs the chain rule in every backward pass).
__global__ void k_variance_to_inv_s(const float* __restrict__ variance, float* __restrict__ inv_s) {
    if (threadIdx.x == 0 && blockIdx.x == 0) inv_s[0] = fminf(fmaxf(expf(10.f * variance[0]), 1e-6f), 1e6f);
}
__global__ void k_variance_chain(const float* __restrict__ g_inv_s, const float* __restrict__ inv_s, float* __restrict__ g_variance) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float v = inv_s[0];
        g_variance[0] = g_inv_s[0] * ((v > 1e-6f && v < 1e6f) ? 10.f * v : 0.f);
    }
}
int variance_to_inv_s(const float* variance, float* inv_s, hipStream_t s) {
    HN_REQUIRE(variance && inv_s, "variance_to_inv_s: null argument");
    hipLaunchKernelGGL(k_variance_to_inv_s, dim3(1), dim3(64), 0, s, variance, inv_s);
    HN_LAUNCH_CHECK();
    return HN_OK;
}
int variance_chain(const float* g_inv_s, const float* inv_s, float* g_variance, hipStream_t s) {
    HN_REQUIRE(g_inv_s && inv_s && g_variance, "variance_chain: null argument");
    hipLaunchKernelGGL(k_variance_chain, dim3(1), dim3(64), 0, s, g_inv_s, inv_s, g_variance);
    HN_LAUNCH_CHECK();
    return HN_OK;
}
int train_loss(const float* color, const float* wsum, const float* gerr, const float* true_rgb, const float* true_mask, int n_rays, float igr_weight,
               float mask_weight, float* terms6, hipStream_t s) {
    HN_REQUIRE(color && wsum && gerr && true_rgb && true_mask && terms6 && n_rays >= 1, "train_loss: bad arguments");
    hipLaunchKernelGGL(k_train_loss, dim3(1), dim3(1024), 0, s, color, wsum, gerr, true_rgb, true_mask, n_rays, igr_weight, mask_weight, terms6);
    HN_LAUNCH_CHECK();
    return HN_OK;
}
int train_loss_bwd(const float* color, const float* wsum, const float* true_rgb, const float* true_mask, int n_rays, const float* terms6, const float* g_loss,
                   float igr_weight, float mask_weight, float* g_color, float* g_wsum, float* g_gerr, hipStream_t s) {
    HN_REQUIRE(color && wsum && true_rgb && true_mask && terms6 && g_loss && g_color && g_wsum && g_gerr && n_rays >= 1, "train_loss_bwd: bad arguments");
    hipLaunchKernelGGL(k_train_loss_bwd, dim3((n_rays + 255) / 256), dim3(256), 0, s, color, wsum, true_rgb, true_mask, n_rays, terms6, g_loss, igr_weight,
                       mask_weight, g_color, g_wsum, g_gerr);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

}  // namespace hn
