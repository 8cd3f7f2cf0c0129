// The pose-only and loss side of a fitting_video window step as a handful of launches (fitting_video.py:285-342,
// utils/renderer_batch.py:318-371).  As torch operators these were ~250 of the step's 366 launches, and the step is bound by its
// launches, not by its kernels (profiles/r03/fit_video_step_*): the stable term ~50 forward + ~90 backward (19 of them the
// layer-by-layer sdf adjoint), the window's loss / regularisers / smoothness ~60 + ~80, torch.inverse and its backward ~15.
//   k_mat3_inverse(_bwd)   torch.inverse(obj_r) of fitting_video.py:284 for [F,3,3] and its adjoint
//   k_stable_pts(_bwd)     the object's vertices taken to the world, pts[:, ::10] (utils/renderer_batch.py:319-321), and the adjoint
//   k_stable_value         everything of get_stable_loss_cross behind the hand SDF: inside sets, the nearest outside vertex of
//                          every inside vertex (the reference's cKDTree query), weights, the value and d value / d sdf
//   k_window_loss(_bwd)    the whole loss of the window in one launch each way (render terms, contact / penetration, joint and
//                          vertex regularisers, smoothness with its sequence-end anchors, the stable term)
#include "hn_common.h"

namespace hn {

__device__ __forceinline__ float wsum64(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ---- 3 x 3 inverse by the adjugate (fitting_video.py:284: Ro = torch.inverse(obj_r)) ----------------------------------------------
__global__ void k_mat3_inverse(const float* __restrict__ R, int n, float* __restrict__ out) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    const float* m = R + 9 * f;
    const float a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], g = m[5], h = m[6], i = m[7], j = m[8];
    const float A = e * j - g * i, B = -(d * j - g * h), C = d * i - e * h;
    const float inv = 1.f / (a * A + b * B + c * C);
    float* o = out + 9 * f;
    o[0] = A * inv;
    o[1] = -(b * j - c * i) * inv;
    o[2] = (b * g - c * e) * inv;
    o[3] = B * inv;
    o[4] = (a * j - c * h) * inv;
    o[5] = -(a * g - c * d) * inv;
    o[6] = C * inv;
    o[7] = -(a * i - b * h) * inv;
    o[8] = (a * e - b * d) * inv;
}
// Y = R^-1:  dL/dR = -Y^T (dL/dY) Y^T
__global__ void k_mat3_inverse_bwd(const float* __restrict__ Y, const float* __restrict__ gY, int n, float* __restrict__ gR) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    const float* y = Y + 9 * f;
    const float* g = gY + 9 * f;
    float t[9];   // t = Y^T g
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) t[3 * r + c] = y[r] * g[c] + y[3 + r] * g[3 + c] + y[6 + r] * g[6 + c];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) gR[9 * f + 3 * r + c] = -(t[3 * r] * y[3 * c] + t[3 * r + 1] * y[3 * c + 1] + t[3 * r + 2] * y[3 * c + 2]);   // (t Y^T)[r][c]
}

// ---- world positions of every `stride`-th object vertex: p_w[f,v] = R_f p[f, stride v] + t_f -------------------------------------
// p0_out [V,3]: frame 0's selected vertices in object coordinates (what the nearest-vertex query of the stable term runs on)
__global__ void k_stable_pts(const float* __restrict__ pts, int n_frames, int n_full, int stride, int V, const float* __restrict__ R,
                             const float* __restrict__ t, float* __restrict__ out, float* __restrict__ p0_out) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_frames * V) return;
    const int f = q / V, v = q % V;
    const float* p = pts + ((size_t)f * n_full + (size_t)v * stride) * 3;
    const float* m = R + 9 * f;
#pragma unroll
    for (int r = 0; r < 3; ++r) out[3 * (size_t)q + r] = (m[3 * r] * p[0] + m[3 * r + 1] * p[1] + m[3 * r + 2] * p[2]) + t[3 * f + r];
    if (f == 0 && p0_out != nullptr) {
#pragma unroll
        for (int r = 0; r < 3; ++r) p0_out[3 * v + r] = p[r];
    }
}
// g_R[f] = sum_v g[f,v] p^T, g_t[f] = sum_v g[f,v]: one block per frame
__global__ __launch_bounds__(256) void k_stable_pts_bwd(const float* __restrict__ pts, int n_full, int stride, int V, const float* __restrict__ g,
                                                        float* __restrict__ gR, float* __restrict__ gt) {
    const int f = blockIdx.x;
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.f;
    for (int v = threadIdx.x; v < V; v += blockDim.x) {
        const float* p = pts + ((size_t)f * n_full + (size_t)v * stride) * 3;
        const float* gv = g + ((size_t)f * V + v) * 3;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[3 * r + c] += gv[r] * p[c];
            acc[9 + r] += gv[r];
        }
    }
    __shared__ float red[12][4];
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const float s = wsum64(acc[k]);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = s;
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        const float s = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
        if (threadIdx.x < 9)
            gR[9 * f + threadIdx.x] = s;
        else
            gt[3 * f + threadIdx.x - 9] = s;
    }
}

// ---- get_stable_loss_cross behind the hand SDF (utils/renderer_batch.py:326-369), one block ---------------------------------------
// sdf [F,V] on the world vertices; p0 [V,3] = pts[0, ::stride] (the reference queries frame 0's vertices, :352-353).  Over the frames
// in which the hand penetrates the object (pen_f: any sdf < 0; in_time of them) and per such frame cid:
//   in_err  = sum over ALL penetrating frames and cid's inside vertices of clip(sdf, 0, .) / ((in_time - 1) n_in[cid])
//   out_err = the same over the nearest 'outside' vertex of every inside vertex of cid, of |clip(sdf, ., 0)|
//   value = sum_cid (in_err + 0.05 out_err) / in_time   (0 unless in_time > 1)
// strict: the reference's 'outside' set -- np.setdiff1d(range(V), mask) applied to the BOOLEAN mask removes vertex 1 if any vertex is
// inside and vertex 0 if any is outside, and keeps the inside vertices; else the complement of the inside set.
// Outputs: value[0], dsdf [F,V] = d value / d sdf (the sets and weights are constants, as under the reference's .cpu() / numpy).
constexpr int STABLE_MAX_F = 8, STABLE_MAX_V = 1024;
constexpr int STABLE_QPB = 16;   // queries per block (4 per wave)
// Many blocks for the nearest-vertex queries (one wave per query: 800 of them in a window of 4 x 200 vertices all inside the hand,
// ~1 us each: 0.35 ms in one block), the block that finishes last does the rest.  nearest [F V] ints and the counter (zero at the
// first launch, left zero by every launch) live in the caller's scratch.
__global__ __launch_bounds__(256) void k_stable_value(const float* __restrict__ sdf, const float* __restrict__ p0, int F, int V, int strict,
                                                      float* __restrict__ value, float* __restrict__ dsdf, int* __restrict__ nearest,
                                                      unsigned* __restrict__ counter) {
    __shared__ int n_in[STABLE_MAX_F], any_out[STABLE_MAX_F];
    __shared__ float red[4];
    __shared__ bool is_last;
    __shared__ unsigned char sel[STABLE_MAX_F * STABLE_MAX_V];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < STABLE_MAX_F) {
        n_in[tid] = 0;
        any_out[tid] = 0;
    }
    __syncthreads();
    for (int q = tid; q < F * V; q += blockDim.x) {
        const int f = q / V;
        if (sdf[q] < 0.f)
            atomicAdd(&n_in[f], 1);
        else
            atomicOr(&any_out[f], 1);
    }
    __syncthreads();
    // nearest candidate of every inside vertex of every penetrating frame (brute force over the V candidates; ties: lowest index)
    for (int k = 0; k < STABLE_QPB / 4; ++k) {
        const int q = blockIdx.x * STABLE_QPB + wave * (STABLE_QPB / 4) + k;
        if (q >= F * V) break;
        const int f = q / V, i = q % V;
        int arg = -1;
        if (n_in[f] > 0 && sdf[q] < 0.f) {
            const float px = p0[3 * i], py = p0[3 * i + 1], pz = p0[3 * i + 2];
            float best = INFINITY;
            for (int j = lane; j < V; j += 64) {
                bool cand;
                if (strict)
                    cand = !((j == 1 && n_in[f] > 0) || (j == 0 && any_out[f] != 0));
                else
                    cand = !(sdf[f * V + j] < 0.f);
                if (!cand) continue;
                const float dx = p0[3 * j] - px, dy = p0[3 * j + 1] - py, dz = p0[3 * j + 2] - pz;
                const float d = dx * dx + dy * dy + dz * dz;
                if (d < best) {
                    best = d;
                    arg = j;
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float ob = __shfl_xor(best, off, 64);
                const int oa = __shfl_xor(arg, off, 64);
                if (oa >= 0 && (arg < 0 || ob < best || (ob == best && oa < arg))) {
                    best = ob;
                    arg = oa;
                }
            }
        }
        if (lane == 0) nearest[q] = arg;
    }
    __threadfence();
    __syncthreads();
    if (tid == 0) is_last = atomicAdd(counter, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    if (tid == 0) *counter = 0u;
    for (int q = tid; q < F * V; q += blockDim.x) sel[q] = 0;
    __syncthreads();
    for (int q = tid; q < F * V; q += blockDim.x) {
        const int a = reinterpret_cast<const volatile int*>(nearest)[q];
        if (a >= 0) sel[(q / V) * V + a] = 1;   // (several queries may name the same vertex: the same byte, the same value)
    }
    __syncthreads();
    int in_time = 0;
    for (int f = 0; f < F; ++f) in_time += n_in[f] > 0 ? 1 : 0;
    // per vertex: Win = sum_cid inside[cid,v] / denom[cid], Wout = sum_cid selected[cid,v] / denom[cid]; pos / neg over the penetrating frames
    float total = 0.f;
    const float inv_time = in_time > 1 ? 1.f / (float)in_time : 0.f;
    for (int v = tid; v < V; v += blockDim.x) {
        float Win = 0.f, Wout = 0.f, pos = 0.f, neg = 0.f;
        for (int f = 0; f < F; ++f) {
            if (n_in[f] <= 0) continue;
            const float denom = fmaxf((float)(in_time - 1) * (float)n_in[f], 1.f);
            const float s = sdf[f * V + v];
            if (s < 0.f) Win += 1.f / denom;
            if (sel[f * V + v]) Wout += 1.f / denom;
            pos += fminf(fmaxf(s, 0.f), 1e7f);
            neg += fabsf(fmaxf(fminf(s, 0.f), -1e7f));
        }
        total += Win * pos + 0.05f * (Wout * neg);
        for (int f = 0; f < F; ++f) {
            float g = 0.f;
            if (n_in[f] > 0) {
                const float s = sdf[f * V + v];
                // clip(s, 0, 1e7) passes the gradient on [0, 1e7]; |clip(s, -1e7, 0)|: -1 for -1e7 <= s < 0, 0 at 0 (abs'(0) = 0)
                if (s >= 0.f && s <= 1e7f) g += Win;
                if (s < 0.f && s >= -1e7f) g -= 0.05f * Wout;
            }
            dsdf[f * V + v] = g * inv_time;
        }
    }
    total = wsum64(total);
    if (lane == 0) red[wave] = total;
    __syncthreads();
    if (tid == 0) value[0] = ((red[0] + red[1]) + (red[2] + red[3])) * inv_time;
}

// ---- the whole loss of a window (fitting_video.py:285-334) -------------------------------------------------------------------------
// loss = 0.5 (colour + 0.5 mask) + 30 contact + 20 penetration + 30 joint + 20 verts + 50 smooth + 100 stable,
//   colour = sum |(c - c*) m| / (F P), mask = mean BCE(clip(w, 1e-3, 1 - 1e-3), m), contact / penetration as in fitting_single,
//   joint = mean_{f,k} |j_fk - jp_fk|, verts = mean_f mean_v |(R_f - Rp_f) v + (t_f - tp_f)|  (pose_loss = mean, :123-126),
//   smooth = mean_{f<F-1,k} |j_{f+1,k} - j_fk| + mean_{f<F-1} mean_v |(R_{f+1} - R_f) v + (t_{f+1} - t_f)|
//            [+ mean_k |j_0k - jp_0k| + verts_0 when the window starts the sequence, else + the same of the last frame when it ends
//            it; not on the very first step: the caller's `anchor`], stable: get_stable_loss_cross (a device scalar or NULL).
// terms10 = {loss, colour, mask, contact, penetration, joint, verts, smooth x 50, stable x 100, 0}; g_joint [F,21,3], gR [F,9], gt
// [F,3]: d (the weighted pose part of the loss) / d (joint_3d, obj_r, obj_t).  Same structure as k_fit_step_loss: per-block
// partial sums, the last block adds them in index order and does the pose part.
struct WindowW {
    float render, contact, penet, joint, verts, smooth, stable;
};
__global__ __launch_bounds__(256) void k_window_loss(const float* __restrict__ color, const float* __restrict__ wsum, const float* __restrict__ true_rgb,
                                                     const float* __restrict__ true_mask, int n_rays, const float* __restrict__ sdf_h,
                                                     const float* __restrict__ sdf_o, int n_samples, const float* __restrict__ joint_3d,
                                                     const float* __restrict__ joint_pred, int F, const float* __restrict__ R, const float* __restrict__ t,
                                                     const float* __restrict__ Rp, const float* __restrict__ tp, const float* __restrict__ verts,
                                                     int n_verts, const float* __restrict__ stable, int anchor, WindowW w, float* __restrict__ partials,
                                                     unsigned* __restrict__ counter, float* __restrict__ pose3, float* __restrict__ sums6,
                                                     float* __restrict__ terms10, float* __restrict__ g_joint, float* __restrict__ gR,
                                                     float* __restrict__ gt, float* __restrict__ pairs_g) {
    __shared__ float red[13][4];
    __shared__ float pair_out[2 * STABLE_MAX_F][13];
    __shared__ bool is_last;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (i < n_rays) {
        const float m = true_mask[i], inv = 1.f / (float)n_rays;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[0] += fabsf((color[3 * (size_t)i + c] - true_rgb[3 * (size_t)i + c]) * m);
        const float ww = fminf(fmaxf(wsum[i], 1e-3f), 1.f - 1e-3f);
        v[0] *= inv;
        v[1] = -(m * logf(ww) + (1.f - m) * logf(1.f - ww)) * inv;
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
        const float s = wsum64(v[k]);
        if (lane == 0) red[k][wave] = s;
    }
    __syncthreads();
    if (threadIdx.x < 6) partials[6 * (size_t)blockIdx.x + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
    __syncthreads();
    // ---- the pose part (does not depend on the render) ---------------------------------------------------------------------------
    // vertex pairs: p < F: (R_p, t_p) against the prediction; p >= F: (R_{q+1}, t_{q+1}) against (R_q, t_q), q = p - F.
    // pairs_g[p] = {mean |e|, mean u v^T (9), mean u (3)} (k_verts_loss).  One pair per BLOCK (block p, beside its share of the sums): as
    // a loop of block 0 over the 2 F - 1 pairs the launch took 48 us, most of it that loop; the last block combines them.
    const int n_pairs = 2 * F - 1;
    {
        for (int p = blockIdx.x; p < n_pairs; p += gridDim.x) {
            const float *Ra, *ta, *Rb, *tb;
            if (p < F) {
                Ra = R + 9 * p; ta = t + 3 * p; Rb = Rp + 9 * p; tb = tp + 3 * p;
            } else {
                const int q = p - F;
                Ra = R + 9 * (q + 1); ta = t + 3 * (q + 1); Rb = R + 9 * q; tb = t + 3 * q;
            }
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
                const float s = wsum64(a13[k]);
                if (lane == 0) red[k][wave] = s;
            }
            __syncthreads();
            if (threadIdx.x < 13) pairs_g[13 * p + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]) / (float)n_verts;
            __syncthreads();
        }
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) is_last = atomicAdd(counter, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    {
        {   // the pairs of all blocks -> LDS
            const volatile float* pg = reinterpret_cast<const volatile float*>(pairs_g);
            for (int q = threadIdx.x; q < 13 * n_pairs; q += blockDim.x) pair_out[q / 13][q % 13] = pg[q];
        }
        __syncthreads();
        // joints: thread (f, k) for f < F, k < 21 (F <= 8: 168 threads)
        const int NJ = 21;
        const bool first = (anchor & 1) != 0, last = !first && (anchor & 2) != 0;
        float jl = 0.f, js = 0.f, ja = 0.f;   // this thread's |j - jp|, |j_{f+1} - j_f|, anchor term
        float gj[3] = {0.f, 0.f, 0.f};
        const int tf = threadIdx.x / NJ, tk = threadIdx.x % NJ;
        const float cj = w.joint / (float)(NJ * F), cs = F > 1 ? w.smooth / (float)(NJ * (F - 1)) : 0.f, ca = w.smooth / (float)NJ;
        if (threadIdx.x < NJ * F) {
            const float* a = joint_3d + (size_t)(tf * NJ + tk) * 3;
            const float* b = joint_pred + (size_t)(tf * NJ + tk) * 3;
            auto unit = [](const float* x, const float* y, float (&u)[3]) {
                const float e0 = x[0] - y[0], e1 = x[1] - y[1], e2 = x[2] - y[2];
                const float n = sqrtf(e0 * e0 + e1 * e1 + e2 * e2);
                const float inv = n > 0.f ? 1.f / n : 0.f;   // torch.norm's subgradient at 0 is 0
                u[0] = e0 * inv;
                u[1] = e1 * inv;
                u[2] = e2 * inv;
                return n;
            };
            float u[3];
            jl = unit(a, b, u);
#pragma unroll
            for (int c = 0; c < 3; ++c) gj[c] += cj * u[c];
            if ((first && tf == 0) || (last && tf == F - 1)) {
                ja = jl;
#pragma unroll
                for (int c = 0; c < 3; ++c) gj[c] += ca * u[c];
            }
            if (tf + 1 < F) {   // |j_{f+1} - j_f|: this thread owns the pair's value; -gradient to j_f
                js = unit(a + NJ * 3, a, u);
#pragma unroll
                for (int c = 0; c < 3; ++c) gj[c] -= cs * u[c];
            }
            if (tf > 0) {       // +gradient of the pair (f - 1, f) to j_f
                unit(a, a - NJ * 3, u);
#pragma unroll
                for (int c = 0; c < 3; ++c) gj[c] += cs * u[c];
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) g_joint[(size_t)(tf * NJ + tk) * 3 + c] = gj[c];
        }
        const float s_jl = wsum64(jl), s_js = wsum64(js), s_ja = wsum64(ja);
        if (lane == 0) {
            red[0][wave] = s_jl;
            red[1][wave] = s_js;
            red[2][wave] = s_ja;
        }
        __syncthreads();
        // pose gradients w.r.t. (R_f, t_f): thread (f, e), e < 12
        if (threadIdx.x < 12 * F) {
            const int f = threadIdx.x / 12, e = threadIdx.x % 12;
            const int col = 1 + e;   // pair_out column of element e of (gR | gt)
            float g = (w.verts / (float)F) * pair_out[f][col];
            if ((first && f == 0) || (last && f == F - 1)) g += w.smooth * pair_out[f][col];
            const float csv = F > 1 ? w.smooth / (float)(F - 1) : 0.f;
            if (f > 0) g += csv * pair_out[F + f - 1][col];          // pair (f - 1, f): this frame is the `a` side
            if (f + 1 < F) g -= csv * pair_out[F + f][col];          // pair (f, f + 1): the `b` side
            if (e < 9)
                gR[9 * f + e] = g;
            else
                gt[3 * f + e - 9] = g;
        }
        if (threadIdx.x == 0) {
            const float joint = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (float)(NJ * F);
            float verts_l = 0.f, sm_v = 0.f;
            for (int f = 0; f < F; ++f) verts_l += pair_out[f][0];
            verts_l /= (float)F;
            for (int q = 0; q + 1 < F; ++q) sm_v += pair_out[F + q][0];
            float smooth = F > 1 ? ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / (float)(NJ * (F - 1)) + sm_v / (float)(F - 1) : 0.f;
            if (first) smooth += ((red[2][0] + red[2][1]) + (red[2][2] + red[2][3])) / (float)NJ + pair_out[0][0];
            if (last) smooth += ((red[2][0] + red[2][1]) + (red[2][2] + red[2][3])) / (float)NJ + pair_out[F - 1][0];
            pose3[0] = joint;
            pose3[1] = verts_l;
            pose3[2] = smooth;
        }
        __syncthreads();
    }
    {   // the six sums over the blocks' slots, in a fixed order
        float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const volatile float* ps = reinterpret_cast<const volatile float*>(partials);
        for (unsigned b = threadIdx.x; b < gridDim.x; b += 256) {
#pragma unroll
            for (int k = 0; k < 6; ++k) acc[k] += ps[6 * (size_t)b + k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float s = wsum64(acc[k]);
            if (lane == 0) red[k][wave] = s;
        }
        __syncthreads();
        if (threadIdx.x < 6) {
            const float s = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
            sums6[threadIdx.x] = s;
            red[threadIdx.x][0] = s;
        }
        if (threadIdx.x == 0) *counter = 0u;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float colour = red[0][0], mask = red[1][0];
        const float contact = red[2][0] / (red[3][0] + 1e-9f), penet = red[4][0] / (red[5][0] + 1e-9f);
        const volatile float* pz = reinterpret_cast<const volatile float*>(pose3);
        const float joint = pz[0], verts_l = pz[1], smooth = pz[2];
        const float st = stable != nullptr ? stable[0] : 0.f;
        terms10[0] = w.render * (colour + 0.5f * mask) + (w.contact * contact + w.penet * penet) + (w.joint * joint + w.verts * verts_l) + w.smooth * smooth +
                     w.stable * st;
        terms10[1] = colour;
        terms10[2] = mask;
        terms10[3] = contact;
        terms10[4] = penet;
        terms10[5] = joint;
        terms10[6] = verts_l;
        terms10[7] = w.smooth * smooth;
        terms10[8] = w.stable * st;
        terms10[9] = 0.f;
    }
}
__global__ __launch_bounds__(256) void k_window_loss_bwd(const float* __restrict__ color, const float* __restrict__ wsum, const float* __restrict__ true_rgb,
                                                         const float* __restrict__ true_mask, int n_rays, const float* __restrict__ sdf_h,
                                                         const float* __restrict__ sdf_o, int n_samples, const float* __restrict__ sums,
                                                         const float* __restrict__ g_loss, WindowW w, const float* __restrict__ g_joint,
                                                         const float* __restrict__ gR, const float* __restrict__ gt, int F, float* __restrict__ g_color,
                                                         float* __restrict__ g_wsum, float* __restrict__ g_sdf_h, float* __restrict__ g_sdf_o,
                                                         float* __restrict__ g_joint_out, float* __restrict__ gR_out, float* __restrict__ gt_out,
                                                         float* __restrict__ g_stable) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float gl = g_loss[0];
    const float g0 = gl * w.render, g1 = gl * w.render * 0.5f, g2 = gl * w.contact, g3 = gl * w.penet;
    if (blockIdx.x == 0) {   // (the pose-side gradients already carry their weights: scaled by the upstream gradient only)
        for (int k = threadIdx.x; k < 63 * F; k += blockDim.x) g_joint_out[k] = gl * g_joint[k];
        for (int k = threadIdx.x; k < 9 * F; k += blockDim.x) gR_out[k] = gl * gR[k];
        for (int k = threadIdx.x; k < 3 * F; k += blockDim.x) gt_out[k] = gl * gt[k];
        if (threadIdx.x == 0 && g_stable != nullptr) g_stable[0] = gl * w.stable;
    }
    if (i < n_rays) {
        const float m = true_mask[i], inv = 1.f / (float)n_rays;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float e = (color[3 * (size_t)i + c] - true_rgb[3 * (size_t)i + c]) * m;
            g_color[3 * (size_t)i + c] = g0 * inv * m * (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f));
        }
        const float wv = wsum[i];
        const float ww = fminf(fmaxf(wv, 1e-3f), 1.f - 1e-3f);
        const bool inside = wv >= 1e-3f && wv <= 1.f - 1e-3f;
        g_wsum[i] = inside ? g1 * inv * (ww - m) / (ww * (1.f - ww)) : 0.f;
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

// ---- host entry points ---------------------------------------------------------------------------------------------------------
int mat3_inverse(const float* R, int n, float* out, hipStream_t s) {
    if (n <= 0) return HN_OK;
    HN_REQUIRE(R != nullptr && out != nullptr, "mat3_inverse: NULL argument");
    hipLaunchKernelGGL(k_mat3_inverse, dim3((n + 63) / 64), dim3(64), 0, s, R, n, out);
    HN_LAUNCH_CHECK();
    return HN_OK;
}
int mat3_inverse_bwd(const float* Y, const float* gY, int n, float* gR, hipStream_t s) {
    if (n <= 0) return HN_OK;
    HN_REQUIRE(Y != nullptr && gY != nullptr && gR != nullptr, "mat3_inverse_bwd: NULL argument");
    hipLaunchKernelGGL(k_mat3_inverse_bwd, dim3((n + 63) / 64), dim3(64), 0, s, Y, gY, n, gR);
    HN_LAUNCH_CHECK();
    return HN_OK;
}
int stable_pts(const float* pts, int n_frames, int n_full, int stride, const float* R, const float* t, float* out, float* p0_out, hipStream_t s) {
    HN_REQUIRE(pts && R && t && out && n_frames >= 1 && n_full >= 1 && stride >= 1, "stable_pts: bad arguments");
    const int V = (n_full + stride - 1) / stride;
    hipLaunchKernelGGL(k_stable_pts, dim3((n_frames * V + 255) / 256), dim3(256), 0, s, pts, n_frames, n_full, stride, V, R, t, out, p0_out);
    HN_LAUNCH_CHECK();
    return HN_OK;
}
int stable_pts_bwd(const float* pts, int n_frames, int n_full, int stride, const float* g, float* gR, float* gt, hipStream_t s) {
    HN_REQUIRE(pts && g && gR && gt && n_frames >= 1 && n_full >= 1 && stride >= 1, "stable_pts_bwd: bad arguments");
    const int V = (n_full + stride - 1) / stride;
    hipLaunchKernelGGL(k_stable_pts_bwd, dim3(n_frames), dim3(256), 0, s, pts, n_full, stride, V, g, gR, gt);
    HN_LAUNCH_CHECK();
    return HN_OK;
}
size_t stable_value_scratch_bytes(int n_frames, int V) { return ((size_t)n_frames * V + 16) * sizeof(int); }
int stable_value(const float* sdf, const float* p0, int n_frames, int V, int strict, float* value, float* dsdf, void* scratch, size_t scratch_bytes,
                 hipStream_t s) {
    HN_REQUIRE(sdf && p0 && value && dsdf && scratch, "stable_value: NULL argument");
    HN_REQUIRE(n_frames >= 1 && n_frames <= STABLE_MAX_F && V >= 1 && V <= STABLE_MAX_V, "stable_value: at most %d frames of %d vertices", STABLE_MAX_F,
               STABLE_MAX_V);
    HN_REQUIRE(scratch_bytes >= stable_value_scratch_bytes(n_frames, V), "stable_value: scratch too small");
    unsigned* counter = reinterpret_cast<unsigned*>(scratch);   // zero when the scratch is first handed over; every launch leaves it zero
    int* nearest = reinterpret_cast<int*>(scratch) + 16;
    hipLaunchKernelGGL(k_stable_value, dim3((n_frames * V + STABLE_QPB - 1) / STABLE_QPB), dim3(256), 0, s, sdf, p0, n_frames, V, strict, value, dsdf, nearest,
                       counter);
    HN_LAUNCH_CHECK();
    return HN_OK;
}
size_t window_loss_scratch_bytes(int n_rays, int n_samples) {
    const int n = n_rays > n_samples ? n_rays : n_samples;
    return ((size_t)((n + 255) / 256 + 1) * 6 + 16 + 2 * STABLE_MAX_F * 13) * sizeof(float);   // counter, pose terms, the blocks' sums, the vertex pairs
}
int window_loss(const float* color, const float* wsum, const float* true_rgb, const float* true_mask, int n_rays, const float* sdf_h, const float* sdf_o,
                int n_samples, const float* joint_3d, const float* joint_pred, int n_frames, const float* R, const float* t, const float* Rp, const float* tp,
                const float* verts, int n_verts, const float* stable, int anchor, const float* w7, void* scratch, size_t scratch_bytes, float* sums6,
                float* terms10, float* g_joint, float* gR, float* gt, hipStream_t s) {
    HN_REQUIRE(color && wsum && true_rgb && true_mask && joint_3d && joint_pred && R && t && Rp && tp && verts && w7 && scratch && sums6 && terms10 && g_joint &&
                   gR && gt,
               "window_loss: NULL argument");
    HN_REQUIRE(n_rays >= 1 && (sdf_h == nullptr) == (sdf_o == nullptr) && n_frames >= 2 && n_frames <= STABLE_MAX_F && n_verts >= 1, "window_loss: bad sizes");
    const int ns = sdf_h != nullptr ? n_samples : 0;
    HN_REQUIRE(scratch_bytes >= window_loss_scratch_bytes(n_rays, ns), "window_loss: scratch too small");
    const int n = n_rays > ns ? n_rays : ns;
    WindowW w{w7[0], w7[1], w7[2], w7[3], w7[4], w7[5], w7[6]};
    hipLaunchKernelGGL(k_window_loss, dim3((n + 255) / 256), dim3(256), 0, s, color, wsum, true_rgb, true_mask, n_rays, sdf_h, sdf_o, ns, joint_3d, joint_pred,
                       n_frames, R, t, Rp, tp, verts, n_verts, stable, anchor, w, reinterpret_cast<float*>(scratch) + 16, reinterpret_cast<unsigned*>(scratch),
                       reinterpret_cast<float*>(scratch) + 4, sums6, terms10, g_joint, gR, gt,
                       reinterpret_cast<float*>(scratch) + 16 + (size_t)((n + 255) / 256 + 1) * 6);
    HN_LAUNCH_CHECK();
    return HN_OK;
}
int window_loss_bwd(const float* color, const float* wsum, const float* true_rgb, const float* true_mask, int n_rays, const float* sdf_h, const float* sdf_o,
                    int n_samples, const float* sums6, const float* g_loss, const float* w7, const float* g_joint, const float* gR, const float* gt,
                    int n_frames, float* g_color, float* g_wsum, float* g_sdf_h, float* g_sdf_o, float* g_joint_out, float* gR_out, float* gt_out,
                    float* g_stable, hipStream_t s) {
    HN_REQUIRE(color && wsum && true_rgb && true_mask && sums6 && g_loss && w7 && g_joint && gR && gt && g_color && g_wsum && g_joint_out && gR_out && gt_out,
               "window_loss_bwd: NULL argument");
    HN_REQUIRE(sdf_h == nullptr || (sdf_o && g_sdf_h && g_sdf_o), "sdf gradients need both fields");
    const int ns = sdf_h != nullptr ? n_samples : 0;
    const int n = n_rays > ns ? n_rays : ns;
    if (n <= 0) return HN_OK;
    WindowW w{w7[0], w7[1], w7[2], w7[3], w7[4], w7[5], w7[6]};
    hipLaunchKernelGGL(k_window_loss_bwd, dim3((n + 255) / 256), dim3(256), 0, s, color, wsum, true_rgb, true_mask, n_rays, sdf_h, sdf_o, ns, sums6, g_loss, w,
                       g_joint, gR, gt, n_frames, g_color, g_wsum, g_sdf_h, g_sdf_o, g_joint_out, gR_out, gt_out, g_stable);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

}  // namespace hn
