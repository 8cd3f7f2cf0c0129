// SDF -> alpha (utils/renderer.py:147-161) and alpha compositing (utils/renderer.py:163-169,
// 512-524) as wavefront scans: one wave per ray, samples interleaved over the 64 lanes
// (element e = lane + 64 m, so every load/store instruction touches 64 consecutive floats),
// an inclusive product scan across the lanes per 64-sample segment and a scalar carry between
// segments.  These kernels are HBM-bound: 24 B (single) / 40 B (dual) per ray-sample.
#include "hn_common.h"

namespace hn {

__device__ __forceinline__ float sigmoid_e(float x) { return 1.f / (1.f + expf(-x)); }

__global__ void k_alpha(const float* __restrict__ sdf, const float* __restrict__ grad, const float* __restrict__ rays_d,
                        const float* __restrict__ dists, int n, int spr, float inv_s, float* __restrict__ alpha,
                        float* __restrict__ c_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
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
    if (lane == 0 && eik_sum != nullptr && grad != nullptr) atomicAdd(eik_sum, eik_total);
}

// two fields: T_k = prod_{j<k} (1 - a_h + 1e-7)(1 - a_o + 1e-7); w_h = a_h T, w_o = a_o T
__global__ __launch_bounds__(256) void k_composite2(const float* __restrict__ ah, const float* __restrict__ rgbh,
                                                    const float* __restrict__ gh, const float* __restrict__ ao,
                                                    const float* __restrict__ rgbo, const float* __restrict__ go,
                                                    int n_rays, int S, float* __restrict__ color,
                                                    float* __restrict__ weight_sum, float* __restrict__ w_hand,
                                                    float* __restrict__ w_obj, float* __restrict__ eik_sum) {
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
    if (lane == 0 && eik_sum != nullptr) {
        if (gh != nullptr) atomicAdd(eik_sum, eh_total);
        if (go != nullptr) atomicAdd(eik_sum + 1, eo_total);
    }
}

// 4 rays per block; at most 8 blocks per CU, grid-stride beyond that
static int composite_grid(int n_rays) {
    const int blocks = (n_rays + 3) / 4;
    return blocks < 2048 ? blocks : 2048;
}

int alpha(const float* sdf, const float* grad, const float* rays_d, const float* dists, int n, int spr, float inv_s,
          float* alpha_out, float* c, hipStream_t s) {
    HN_REQUIRE(spr > 0, "samples_per_ray must be positive");
    if (n == 0) return HN_OK;
    hipLaunchKernelGGL(k_alpha, dim3((n + 255) / 256), dim3(256), 0, s, sdf, grad, rays_d, dists, n, spr, inv_s,
                       alpha_out, c);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

int composite1(const float* alpha_in, const float* c, const float* rgb, const float* grad, int n_rays, int S,
               float* color, float* weights, float* weight_sum, float* weight_max, float* eik_sum, hipStream_t s) {
    HN_REQUIRE(S >= 1, "S must be positive");
    if (n_rays == 0) return HN_OK;
    hipLaunchKernelGGL(k_composite1, dim3(composite_grid(n_rays)), dim3(256), 0, s, alpha_in, c, rgb, grad, n_rays, S, color,
                       weights, weight_sum, weight_max, eik_sum);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

int composite2(const float* ah, const float* rgbh, const float* gh, const float* ao, const float* rgbo, const float* go,
               int n_rays, int S, float* color, float* weight_sum, float* w_hand, float* w_obj, float* eik_sum,
               hipStream_t s) {
    HN_REQUIRE(S >= 1, "S must be positive");
    if (n_rays == 0) return HN_OK;
    hipLaunchKernelGGL(k_composite2, dim3(composite_grid(n_rays)), dim3(256), 0, s, ah, rgbh, gh, ao, rgbo, go, n_rays, S,
                       color, weight_sum, w_hand, w_obj, eik_sum);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

}  // namespace hn
