// Rays, depth sampling, hierarchical (importance) sampling.
//
// The sampling kernels are deliberately ONE THREAD PER RAY with strictly sequential
// prefix products / prefix sums: the reference builds its cdf with torch.cumprod and
// torch.cumsum (sequential on its CPU path) and inverts it with searchsorted; sample
// indices have to agree bit-for-bit, so the association order is kept (hard
// part 3).  The work is O(k) per ray on a few hundred bytes: launch-latency bound, never
// a bandwidth problem (rows are read once, L1-resident per lane).
#include <atomic>

#include "hn_common.h"

namespace hn {

// ---- _xy_to_ray_bundle (utils/utils.py:31-115) ------------------------------------------------
// PyTorch3D convention (third-party, restated: X_view = ((x-px) z/fx, (y-py) z/fy, z),
// X_world = (X_view - T) R^T).
__global__ void k_ray_gen(const float* __restrict__ xy, const float* __restrict__ R, const float* __restrict__ T,
                          const float* __restrict__ focal, const float* __restrict__ principal, int n_cams,
                          int rays_per_cam, float* __restrict__ rays_o, float* __restrict__ rays_d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_cams * rays_per_cam) return;
    const int cam = i / rays_per_cam;
    const float* Rc = R + 9 * cam;
    const float* Tc = T + 3 * cam;
    const float fx = focal[2 * cam], fy = focal[2 * cam + 1];
    const float px = principal[2 * cam], py = principal[2 * cam + 1];
    const float x = xy[2 * i], y = xy[2 * i + 1];
    float pw[2][3];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float z = (float)(k + 1);
        const float v[3] = {(x - px) * z / fx - Tc[0], (y - py) * z / fy - Tc[1], z - Tc[2]};
        // (X_view - T) @ R^T : out_j = sum_i v_i R[j][i]
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) pw[k][jj] = v[0] * Rc[3 * jj] + v[1] * Rc[3 * jj + 1] + v[2] * Rc[3 * jj + 2];
    }
    float d[3] = {pw[1][0] - pw[0][0], pw[1][1] - pw[0][1], pw[1][2] - pw[0][2]};
    const float nrm = fmaxf(sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), 1e-12f);   // F.normalize eps
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) {
        d[jj] /= nrm;
        rays_d[3 * i + jj] = d[jj];
        rays_o[3 * i + jj] = pw[0][jj] - d[jj];
    }
}

// ---- convert_obj_to_local (utils/renderer.py:180-188) -----------------------------------------
// tr != 0: `Ro` holds the matrix whose TRANSPOSE is the rotation to apply (the caller hands over obj_r where the reference passes
// obj_r.T, fitting_single.py:250): the same products in the same order, read through the transposed index
__global__ void k_obj_local_fwd(const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ Ro,
                                const float* __restrict__ To, int n, int rays_per_frame, float* __restrict__ o_out,
                                float* __restrict__ d_out, int tr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int f = i / rays_per_frame;
    const float* R = Ro + 9 * f;
    const float* T = To + 3 * f;
    const float a[3] = {o[3 * i] - T[0], o[3 * i + 1] - T[1], o[3 * i + 2] - T[2]};
    const float b[3] = {d[3 * i], d[3 * i + 1], d[3 * i + 2]};
    const int sr = tr ? 1 : 3, sc = tr ? 3 : 1;   // element (r, c) of the rotation = R[sr r + sc c]
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        o_out[3 * i + r] = R[sr * r] * a[0] + R[sr * r + sc] * a[1] + R[sr * r + 2 * sc] * a[2];
        d_out[3 * i + r] = R[sr * r] * b[0] + R[sr * r + sc] * b[1] + R[sr * r + 2 * sc] * b[2];
    }
}

// one block per frame: dRo = sum go' (o-To)^T + gd' d^T ; dTo = -Ro^T sum go' ; do = Ro^T go' ; dd = Ro^T gd'
__global__ __launch_bounds__(256) void k_obj_local_bwd(const float* __restrict__ o, const float* __restrict__ d,
                                                       const float* __restrict__ Ro, const float* __restrict__ To,
                                                       const float* __restrict__ go_out,
                                                       const float* __restrict__ gd_out, int rays_per_frame,
                                                       float* __restrict__ g_o, float* __restrict__ g_d,
                                                       float* __restrict__ g_Ro, float* __restrict__ g_To) {
    const int f = blockIdx.x;
    const float* R = Ro + 9 * f;
    const float* T = To + 3 * f;
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.f;
    for (int p = threadIdx.x; p < rays_per_frame; p += blockDim.x) {
        const int i = f * rays_per_frame + p;
        const float a[3] = {o[3 * i] - T[0], o[3 * i + 1] - T[1], o[3 * i + 2] - T[2]};
        const float b[3] = {d[3 * i], d[3 * i + 1], d[3 * i + 2]};
        const float go[3] = {go_out[3 * i], go_out[3 * i + 1], go_out[3 * i + 2]};
        const float gd[3] = {gd_out[3 * i], gd_out[3 * i + 1], gd_out[3 * i + 2]};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[3 * r + c] += go[r] * a[c] + gd[r] * b[c];
            acc[9 + r] += go[r];
        }
        if (g_o != nullptr) {
#pragma unroll
            for (int c = 0; c < 3; ++c) g_o[3 * i + c] = R[c] * go[0] + R[3 + c] * go[1] + R[6 + c] * go[2];
        }
        if (g_d != nullptr) {
#pragma unroll
            for (int c = 0; c < 3; ++c) g_d[3 * i + c] = R[c] * gd[0] + R[3 + c] * gd[1] + R[6 + c] * gd[2];
        }
    }
    __shared__ float red[4][12];
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        float v = acc[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        const int k = threadIdx.x;
        red[0][k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
    }
    __syncthreads();
    if (threadIdx.x < 9) g_Ro[9 * f + threadIdx.x] = red[0][threadIdx.x];
    if (threadIdx.x < 3) {
        const int c = threadIdx.x;
        g_To[3 * f + c] = -(R[c] * red[0][9] + R[3 + c] * red[0][10] + R[6 + c] * red[0][11]);
    }
}

// ---- coarse depths (utils/renderer.py:204-212) ------------------------------------------------
// near, span = far - near and sample_dist = (far - near)/n are rounded from double on the host,
// the way Python floats meet float32 tensors in the reference.
__global__ void k_coarse_z(const float* __restrict__ t_rand, int n_rays, int n, float near, float span,
                           float sample_dist, float* __restrict__ z) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rays * n) return;
    const int b = i / n, k = i % n;
    // torch.linspace(0, 1, n): step = 1/(n-1); first half start + k*step, second half end - (n-1-k)*step
    const float step = 1.f / (float)(n - 1);
    const float lin = (k < n / 2) ? (float)k * step : 1.f - (float)(n - 1 - k) * step;
    z[i] = (near + span * lin) + (t_rand[b] - 0.5f) * sample_dist;
}

// The opening of the two-field render as ONE launch (it was obj_local_fwd + coarse_z + a column copy + a device copy, four
// dependent launches in front of the first field kernel): the object-local rays (k_obj_local_fwd's statements), the shared
// coarse depths (k_coarse_z's statements) written to both sampling tracks and to columns 0 .. n-1 of the [n_rays, S] list
// that collects every round's depths.
__global__ void k_dual_prologue(const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ Ro, const float* __restrict__ To,
                                int n_rays, int rays_per_frame, float* __restrict__ o_out, float* __restrict__ d_out,
                                const float* __restrict__ t_rand, int n, float near, float span, float sample_dist, float* __restrict__ z_hand,
                                float* __restrict__ z_obj, float* __restrict__ zcat, int S, int tr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (Ro != nullptr && i < n_rays) {   // (NULL: the object-local rays are made elsewhere -- on the object branch's own stream)
        const int f = i / rays_per_frame;
        const float* R = Ro + 9 * f;
        const float* T = To + 3 * f;
        const float a[3] = {o[3 * i] - T[0], o[3 * i + 1] - T[1], o[3 * i + 2] - T[2]};
        const float b[3] = {d[3 * i], d[3 * i + 1], d[3 * i + 2]};
        const int sr = tr ? 1 : 3, sc = tr ? 3 : 1;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            o_out[3 * i + r] = R[sr * r] * a[0] + R[sr * r + sc] * a[1] + R[sr * r + 2 * sc] * a[2];
            d_out[3 * i + r] = R[sr * r] * b[0] + R[sr * r + sc] * b[1] + R[sr * r + 2 * sc] * b[2];
        }
    }
    if (i >= n_rays * n) return;
    const int b = i / n, k = i % n;
    const float step = 1.f / (float)(n - 1);
    const float lin = (k < n / 2) ? (float)k * step : 1.f - (float)(n - 1 - k) * step;
    const float zv = (near + span * lin) + (t_rand[b] - 0.5f) * sample_dist;
    z_hand[i] = zv;
    if (z_obj != nullptr) z_obj[i] = zv;
    zcat[(size_t)b * S + k] = zv;
}

// ---- sample positions (utils/renderer.py:216 / 119-123) -----------------------------------------
__global__ void k_sample_points(const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ z,
                                int n_rays, int n, int mid, float sample_dist, float* __restrict__ pts,
                                float* __restrict__ dists) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rays * n) return;
    const int b = i / n, k = i % n;
    float t = z[i];
    if (mid) {
        const float dist = (k + 1 < n) ? z[i + 1] - t : sample_dist;
        dists[i] = dist;
        t = t + dist * 0.5f;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) pts[3 * (size_t)i + c] = o[3 * b + c] + d[3 * b + c] * t;
}

// The same with V consecutive samples per thread when n % V == 0: 16-byte loads of z (+ the next depth), 16-byte stores of
// pts (12 V consecutive bytes) and of dists, instead of 3 + 1 scalar stores at a 12-byte stride.
// (tried and lost: streaming `nt` stores 0.36 of the HBM peak, 8 samples per thread 0.45 - 0.51, an output-major form
// with perfectly coalesced stores but six integer divisions per thread 0.33 - 0.40)
template <int V>   // V consecutive samples per thread (4 or 8), n % V == 0
__global__ void k_sample_points_v(const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ z,
                                  int n_rays, int n, int mid, float sample_dist, float* __restrict__ pts,
                                  float* __restrict__ dists) {
    const unsigned q = blockIdx.x * blockDim.x + threadIdx.x;   // group of V samples (32-bit index arithmetic: one
    const unsigned total = (unsigned)n_rays * (unsigned)n / V;   // unsigned division per thread; the 64-bit form was
    if (q >= total) return;                                      // a third of the kernel's time)
    const unsigned i = q * V;
    const int b = (int)(i / (unsigned)n), k = (int)(i - (unsigned)b * (unsigned)n);
    float t[V];
#pragma unroll
    for (int v = 0; v < V / 4; ++v) {
        const float4 zz = reinterpret_cast<const float4*>(z)[q * (V / 4) + v];
        t[4 * v] = zz.x;
        t[4 * v + 1] = zz.y;
        t[4 * v + 2] = zz.z;
        t[4 * v + 3] = zz.w;
    }
    if (mid) {
        float dd[V];
#pragma unroll
        for (int j = 0; j + 1 < V; ++j) dd[j] = t[j + 1] - t[j];
        dd[V - 1] = (k + V < n) ? z[i + V] - t[V - 1] : sample_dist;
#pragma unroll
        for (int v = 0; v < V / 4; ++v)
            reinterpret_cast<float4*>(dists)[q * (V / 4) + v] = make_float4(dd[4 * v], dd[4 * v + 1], dd[4 * v + 2], dd[4 * v + 3]);
#pragma unroll
        for (int j = 0; j < V; ++j) t[j] = t[j] + dd[j] * 0.5f;
    }
    const float ox = o[3 * b], oy = o[3 * b + 1], oz = o[3 * b + 2], dx = d[3 * b], dy = d[3 * b + 1], dz = d[3 * b + 2];
    float p[3 * V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        p[3 * j] = ox + dx * t[j];
        p[3 * j + 1] = oy + dy * t[j];
        p[3 * j + 2] = oz + dz * t[j];
    }
    float4* out = reinterpret_cast<float4*>(pts) + (3 * V / 4) * q;
#pragma unroll
    for (int v = 0; v < 3 * V / 4; ++v) out[v] = make_float4(p[4 * v], p[4 * v + 1], p[4 * v + 2], p[4 * v + 3]);
}

// 4 samples per thread with the 48 bytes of points a thread produces TRANSPOSED through a wave-private LDS region, so that every
// store instruction of a wave writes 1 KiB of consecutive bytes (as 3 float4 per lane at a 48-byte stride each instruction
// touches every cache line of the wave's 3 KiB and fills a third of it).  No barrier: a wave's LDS operations complete in order.
__global__ __launch_bounds__(256) void k_sample_points_t(const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ z,
                                                         int n_rays, int n, int mid, float sample_dist, float* __restrict__ pts,
                                                         float* __restrict__ dists) {
    __shared__ float4 stage[4][192];
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned q0 = blockIdx.x * 256u + wave * 64u;   // the wave's first group of 4 samples
    const unsigned q = q0 + lane;
    const unsigned total = (unsigned)n_rays * (unsigned)n / 4u;
    const bool ok = q < total;
    if (ok) {
        const unsigned i = q * 4u;
        const int b = (int)(i / (unsigned)n), k = (int)(i - (unsigned)b * (unsigned)n);
        const float4 zz = reinterpret_cast<const float4*>(z)[q];
        float t[4] = {zz.x, zz.y, zz.z, zz.w};
        if (mid) {
            float dd[4];
            dd[0] = t[1] - t[0];
            dd[1] = t[2] - t[1];
            dd[2] = t[3] - t[2];
            dd[3] = (k + 4 < n) ? z[i + 4] - t[3] : sample_dist;
            reinterpret_cast<float4*>(dists)[q] = make_float4(dd[0], dd[1], dd[2], dd[3]);
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = t[j] + dd[j] * 0.5f;
        }
        const float ox = o[3 * b], oy = o[3 * b + 1], oz = o[3 * b + 2], dx = d[3 * b], dy = d[3 * b + 1], dz = d[3 * b + 2];
        float p[12];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            p[3 * j] = ox + dx * t[j];
            p[3 * j + 1] = oy + dy * t[j];
            p[3 * j + 2] = oz + dz * t[j];
        }
#pragma unroll
        for (int v = 0; v < 3; ++v) stage[wave][3 * lane + v] = make_float4(p[4 * v], p[4 * v + 1], p[4 * v + 2], p[4 * v + 3]);
    }
    __builtin_amdgcn_wave_barrier();
    if (q0 >= total) return;
    const unsigned have = (total - q0 < 64u ? total - q0 : 64u) * 3u;   // float4s this wave produced
    float4* out = reinterpret_cast<float4*>(pts) + 3 * (size_t)q0;
#pragma unroll
    for (int v = 0; v < 3; ++v) {
        const unsigned f = v * 64u + lane;
        if (f < have) out[f] = stage[wave][f];
    }
}

// adjoint of k_sample_points w.r.t. the rays (the depths are sampled under no_grad): one wave per ray
//   g_o = sum_k g_pts[k],  g_d = sum_k t_k g_pts[k]   (t_k = z_k, or the section mid-point)
__global__ __launch_bounds__(256) void k_sample_points_bwd(const float* __restrict__ z, const float* __restrict__ g_pts,
                                                           int n_rays, int n, int mid, float sample_dist,
                                                           float* __restrict__ g_o, float* __restrict__ g_d) {
    const int lane = threadIdx.x & 63;
    const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = lane; k < n; k += 64) {
        const size_t i = (size_t)ray * n + k;
        float t = z[i];
        if (mid) t = t + ((k + 1 < n) ? z[i + 1] - t : sample_dist) * 0.5f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float g = g_pts[3 * i + c];
            acc[c] += g;
            acc[3 + c] += t * g;
        }
    }
#pragma unroll
    for (int c = 0; c < 6; ++c)
        for (int off = 32; off > 0; off >>= 1) acc[c] += __shfl_xor(acc[c], off, 64);
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            g_o[3 * ray + c] = acc[c];
            g_d[3 * ray + c] = acc[3 + c];
        }
    }
}

// The object branch behind its adjoint kernel as ONE launch (it was sample_points_bwd + an add + obj_local_bwd): one wave per ray;
//   go = sum_k g_pts[k],  gd = sum_k t_k g_pts[k] + gd_alpha + gd_colour   (d loss / d the object-local ray),
// then the adjoint of o' = Ro (o - To), d' = Ro d:  g_Ro += go (o - To)^T + gd d^T,  g_To -= Ro^T go  (atomics: the caller zeroes
// g_Ro [F,3,3] and g_To [F,3]);  g_o = Ro^T go, g_d = Ro^T gd when the world rays' gradients are wanted.
__global__ __launch_bounds__(256) void k_obj_rays_bwd(const float* __restrict__ z, const float* __restrict__ g_pts, int n_rays, int n, float sample_dist,
                                                      const float* __restrict__ gd_alpha, const float* __restrict__ gd_colour,
                                                      const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ Ro,
                                                      const float* __restrict__ To, int rays_per_frame, float* __restrict__ g_o,
                                                      float* __restrict__ g_d, float* __restrict__ g_Ro, float* __restrict__ g_To, int tr,
                                                      float* __restrict__ part, unsigned* __restrict__ counter, const float* __restrict__ gd_alpha_s,
                                                      const float* __restrict__ gd_colour_s, int finalize) {
    const int lane = threadIdx.x & 63;
    const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool active = ray < n_rays;
    if (!active && (part == nullptr || !finalize)) return;
    if (active) {
    float acc[12] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = lane; k < n; k += 64) {
        const size_t i = (size_t)ray * n + k;
        float t = z[i];
        t = t + ((k + 1 < n) ? z[i + 1] - t : sample_dist) * 0.5f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float g = g_pts[3 * i + c];
            acc[c] += g;
            acc[3 + c] += t * g;
            if (gd_alpha_s != nullptr) acc[6 + c] += gd_alpha_s[3 * i + c];   // the alpha stage's d loss / d rays_d, per sample
            if (gd_colour_s != nullptr) acc[9 + c] += gd_colour_s[3 * i + c];   // the colour network's
        }
    }
#pragma unroll
    for (int c = 0; c < 12; ++c)
        for (int off = 32; off > 0; off >>= 1) acc[c] += __shfl_xor(acc[c], off, 64);
    const int f = ray / rays_per_frame;
    const float* R = Ro + 9 * f;
    const float* T = To + 3 * f;
    const float go[3] = {acc[0], acc[1], acc[2]};
    float gd[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
        gd[c] = acc[3 + c] + (gd_alpha_s != nullptr ? acc[6 + c] : gd_alpha[3 * ray + c]) +
                (gd_colour_s != nullptr ? acc[9 + c] : (gd_colour != nullptr ? gd_colour[3 * ray + c] : 0.f));
    // lanes 0..8: an element of g_Ro; 9..11: of g_To; 12..14 / 15..17: of g_o / g_d.  tr: `Ro` (and g_Ro) hold the transpose of
    // the rotation that was applied (k_obj_local_fwd): element (r, c) of the rotation is R[sr r + sc c]
    const int sr = tr ? 1 : 3, sc = tr ? 3 : 1;
    // g_Ro / g_To: with `part` (the two-field render's backward pass) every ray writes its 12 addends to part [n_rays][12] and the
    // LAST block to finish sums them per frame in ray order -- a fixed order: the same bits in every run, where float atomics
    // gave the same value only to rounding (and a fit's pose a different last bit per run)
    if (lane < 9) {
        const int r = lane / 3, c = lane % 3;
        const float v = go[r] * (o[3 * ray + c] - T[c]) + gd[r] * d[3 * ray + c];
        if (part != nullptr)
            part[(size_t)ray * 12 + sr * r + sc * c] = v;
        else
            atomicAdd(g_Ro + 9 * f + sr * r + sc * c, v);
    } else if (lane < 12) {
        const int c = lane - 9;
        const float v = -(R[sc * c] * go[0] + R[sr + sc * c] * go[1] + R[2 * sr + sc * c] * go[2]);
        if (part != nullptr)
            part[(size_t)ray * 12 + 9 + c] = v;
        else
            atomicAdd(g_To + 3 * f + c, v);
    } else if (lane < 15 && g_o != nullptr) {
        const int c = lane - 12;
        g_o[3 * ray + c] = R[sc * c] * go[0] + R[sr + sc * c] * go[1] + R[2 * sr + sc * c] * go[2];
    } else if (lane >= 15 && lane < 18 && g_d != nullptr) {
        const int c = lane - 15;
        g_d[3 * ray + c] = R[sc * c] * gd[0] + R[sr + sc * c] * gd[1] + R[2 * sr + sc * c] * gd[2];
    }
    }   // active
    if (part == nullptr || !finalize) return;   // (finalize = 0: k_obj_rays_finalize adds the rows, a block per frame)
    // last block finalises (a counter that is zero before the launch and zero again after it)
    __shared__ unsigned is_last;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned done = atomicAdd(counter, 1u) + 1u;
        is_last = done == gridDim.x ? 1u : 0u;
        if (is_last) *counter = 0u;
    }
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    // one wave per (frame, element): lane l adds the frame's rays l, l + 64, ... in that order, then the lanes fold by the fixed
    // xor tree -- the same association in every run
    const int n_frames = (n_rays + rays_per_frame - 1) / rays_per_frame;
    const int wv = threadIdx.x >> 6;
    for (int t = wv; t < n_frames * 12; t += (int)(blockDim.x >> 6)) {
        const int fr = t / 12, e = t % 12;
        const int r0 = fr * rays_per_frame, r1 = r0 + rays_per_frame < n_rays ? r0 + rays_per_frame : n_rays;
        const volatile float* p = part;
        float acc = 0.f;
        for (int r = r0 + lane; r < r1; r += 64) acc += p[(size_t)r * 12 + e];
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (lane == 0) {
            if (e < 9)
                g_Ro[9 * fr + e] = acc;
            else
                g_To[3 * fr + e - 9] = acc;
        }
    }
}

// The per-frame sums of k_obj_rays_bwd's rows with ONE BLOCK PER FRAME (several frames side by side: the last block of that kernel walks
// 12 sums per frame on its four waves -- 24 rounds of dependent loads at 8 frames, 0.1 ms on the object's critical chain of a step).  The same
// association: lane l adds the frame's rays l, l + 64, .. in that order, then the xor tree.
__global__ __launch_bounds__(256) void k_obj_rays_finalize(const float* __restrict__ part, int n_rays, int rays_per_frame, float* __restrict__ g_Ro,
                                                           float* __restrict__ g_To) {
    const int fr = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r0 = fr * rays_per_frame, r1 = r0 + rays_per_frame < n_rays ? r0 + rays_per_frame : n_rays;
    for (int e = wv; e < 12; e += 4) {
        float acc = 0.f;
        for (int r = r0 + lane; r < r1; r += 64) acc += part[(size_t)r * 12 + e];
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (lane == 0) {
            if (e < 9)
                g_Ro[9 * fr + e] = acc;
            else
                g_To[3 * fr + e - 9] = acc;
        }
    }
}

// ---- up_sample + sample_pdf(det=True) (utils/renderer.py:60-86, 10-37) -------------------------
constexpr int UPS_MAX_K = 256;          // the wave / tiled forms (their LDS rows)
constexpr int UPS_DIRECT_MAX_K = 640;   // k_upsample_direct: a cdf row per lane in LDS, 256 k bytes per wave: 160 KB at k = 640
__device__ __forceinline__ float sigmoid_acc(float x) { return 1.f / (1.f + expf(-x)); }

// One section of up_sample's pass 1 (utils/renderer.py:68-84 + sample_pdf's weights + 1e-5): the same operations in the same
// order in every kernel below.
struct UpsState {
    float prev_cos = 0.f, T = 1.f, sum = 0.f, z0, s0;
    __device__ __forceinline__ float step(float z1, float s1, float inv_s) {
        const float mid_sdf = (s0 + s1) * 0.5f;
        const float cosv = (s1 - s0) / (z1 - z0 + 1e-5f);
        float c = fminf(prev_cos, cosv);
        c = fminf(fmaxf(c, -1e3f), 0.f);
        prev_cos = cosv;
        const float dist = z1 - z0;
        const float prev_cdf = sigmoid_acc((mid_sdf - c * dist * 0.5f) * inv_s);
        const float next_cdf = sigmoid_acc((mid_sdf + c * dist * 0.5f) * inv_s);
        const float alpha = (prev_cdf - next_cdf + 1e-5f) / (prev_cdf + 1e-5f);
        const float w = alpha * T + 1e-5f;           // weights + 1e-5 (sample_pdf)
        T = T * (1.f - alpha + 1e-7f);
        sum += w;
        z0 = z1;
        s0 = s1;
        return w;
    }
};
constexpr int UPS_BLK = 8;   // row entries fetched ahead of their use (LDS / global latency is paid once per block, not per entry)

__global__ __launch_bounds__(64) void k_upsample_direct(const float* __restrict__ z, const float* __restrict__ sdf, int n_rays,
                                                 int k, int n_new, float inv_s, float* __restrict__ z_new,
                                                 int64_t* __restrict__ inds_out) {
    extern __shared__ float lds[];   // cdf[k][64]
    const int lane = threadIdx.x;
    const int ray = blockIdx.x * 64 + lane;
    if (ray >= n_rays) return;
    const float* zr = z + (size_t)ray * k;
    const float* sr = sdf + (size_t)ray * k;
    // pass 1: section weights, sequential transmittance (torch.cumprod order)
    UpsState st;
    st.z0 = zr[0];
    st.s0 = sr[0];
    for (int i0 = 1; i0 < k; i0 += UPS_BLK) {
        float zb[UPS_BLK], sb[UPS_BLK];
#pragma unroll
        for (int u = 0; u < UPS_BLK; ++u) {
            const int i = i0 + u < k ? i0 + u : k - 1;
            zb[u] = zr[i];
            sb[u] = sr[i];
        }
#pragma unroll
        for (int u = 0; u < UPS_BLK; ++u)
            if (i0 + u < k) lds[(i0 + u) * 64 + lane] = st.step(zb[u], sb[u], inv_s);
    }
    const float sum = st.sum;
    // pass 2: cdf = [0, cumsum(w / sum)]
    lds[lane] = 0.f;
    float run = 0.f;
    for (int i0 = 1; i0 < k; i0 += UPS_BLK) {
        float wb[UPS_BLK];
#pragma unroll
        for (int u = 0; u < UPS_BLK; ++u) wb[u] = lds[(i0 + u < k ? i0 + u : k - 1) * 64 + lane];
#pragma unroll
        for (int u = 0; u < UPS_BLK; ++u)
            if (i0 + u < k) {
                run += wb[u] / sum;
                lds[(i0 + u) * 64 + lane] = run;
            }
    }
    // pass 3: invert at u = linspace(0.5/n, 1-0.5/n, n)
    const float u_start = (float)(0.0 + 0.5 / (double)n_new), u_end = (float)(1.0 - 0.5 / (double)n_new);   // (torch.linspace of Python doubles: one rounding each)
    const float u_step = n_new > 1 ? (u_end - u_start) / (float)(n_new - 1) : 0.f;   // linspace(steps=1) = [start]
    int ptr = 0;   // number of cdf entries <= u (searchsorted right=True); cdf and u are both non-decreasing
    for (int jj = 0; jj < n_new; ++jj) {
        const float u = (jj < n_new / 2) ? u_start + (float)jj * u_step : u_end - (float)(n_new - 1 - jj) * u_step;
        for (;;) {   // four entries per round trip (the cdf is non-decreasing: the entries <= u are a prefix)
            int adv = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) adv += (ptr + q < k && lds[(ptr + q < k ? ptr + q : k - 1) * 64 + lane] <= u) ? 1 : 0;
            ptr += adv;
            if (adv < 4) break;
        }
        const int below = ptr - 1 > 0 ? ptr - 1 : 0;
        const int above = ptr < k - 1 ? ptr : k - 1;
        const float c_lo = lds[below * 64 + lane], c_hi = lds[above * 64 + lane];
        const float b_lo = zr[below], b_hi = zr[above];
        float denom = c_hi - c_lo;
        denom = denom < 1e-5f ? 1.f : denom;
        const float t = (u - c_lo) / denom;
        z_new[(size_t)ray * n_new + jj] = b_lo + t * (b_hi - b_lo);
        if (inds_out != nullptr) inds_out[(size_t)ray * n_new + jj] = ptr;
    }
}

// Thread per ray with only the weight / cdf row of every ray resident in LDS: the depth and sdf rows pass through an 8-column
// staging tile (read from global memory with consecutive lanes on consecutive 32-byte row segments, the next tile requested
// before the current one is consumed), and the two depths pass 3 needs per new sample come from global memory again (L2).
// 64 x (k + 1) + 2 x 64 x 9 floats per wave instead of 2 x 64 x (k + 1): 7 waves per CU instead of 4 at k = 64 -- the kernel is
// a chain of ~9 000 dependent VALU instructions per wave (four IEEE divisions and two expf per section) and is bound by how
// many such chains a SIMD can interleave.  Same operations in the same order as the kernels above: identical results.
__global__ __launch_bounds__(64) void k_upsample_tiled(const float* __restrict__ z, const float* __restrict__ sdf, int n_rays, int k,
                                                       int n_new, float inv_s, float* __restrict__ z_new, int64_t* __restrict__ inds_out) {
    extern __shared__ float lds[];   // cs[64][k+1] | zc[64][9] | sc[64][9]
    const int lane = threadIdx.x;
    const int pitch = k + 1;
    float* cs = lds;
    float* zc = lds + 64 * pitch;
    float* sc = zc + 64 * 9;
    const int ray0 = blockIdx.x * 64;
    const int nvalid = n_rays - ray0 < 64 ? n_rays - ray0 : 64;
    const int ray = ray0 + lane;
    const bool mine = lane < nvalid;
    float* cr = cs + lane * pitch;
    float za[8], sa[8], zb[8], sb[8];
    auto fetch = [&](int c0, float(&zt)[8], float(&stt)[8]) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = lane + 64 * it, row = idx >> 3, col = c0 + (idx & 7);
            const bool ok = row < nvalid && col < k;
            const size_t g = (size_t)(ray0 + (ok ? row : 0)) * k + (ok ? col : 0);
            zt[it] = z[g];
            stt[it] = sdf[g];
        }
    };
    auto put = [&](const float(&zt)[8], const float(&stt)[8]) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = lane + 64 * it;
            zc[(idx >> 3) * 9 + (idx & 7)] = zt[it];
            sc[(idx >> 3) * 9 + (idx & 7)] = stt[it];
        }
    };
    UpsState st;
    fetch(0, za, sa);
    for (int c0 = 0; c0 < k; c0 += 8) {
        put(za, sa);
        __builtin_amdgcn_wave_barrier();   // (one wave per block: LDS operations complete in order)
        if (c0 + 8 < k) fetch(c0 + 8, zb, sb);
        if (mine) {
            float zv[8], sv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                zv[u] = zc[lane * 9 + u];
                sv[u] = sc[lane * 9 + u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int col = c0 + u;
                if (col == 0) {
                    st.z0 = zv[u];
                    st.s0 = sv[u];
                } else if (col < k) {
                    cr[col] = st.step(zv[u], sv[u], inv_s);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            za[u] = zb[u];
            sa[u] = sb[u];
        }
    }
    if (!mine) return;
    const float sum = st.sum;
    // pass 2: cdf = [0, cumsum(w / sum)]
    cr[0] = 0.f;
    float run = 0.f;
    for (int i0 = 1; i0 < k; i0 += UPS_BLK) {
        float wb[UPS_BLK];
#pragma unroll
        for (int u = 0; u < UPS_BLK; ++u) wb[u] = cr[i0 + u < k ? i0 + u : k - 1];
#pragma unroll
        for (int u = 0; u < UPS_BLK; ++u)
            if (i0 + u < k) {
                run += wb[u] / sum;
                cr[i0 + u] = run;
            }
    }
    // pass 3: invert at u = linspace(0.5/n, 1-0.5/n, n)
    const float* zr = z + (size_t)ray * k;
    const float u_start = (float)(0.0 + 0.5 / (double)n_new), u_end = (float)(1.0 - 0.5 / (double)n_new);   // (torch.linspace of Python doubles: one rounding each)
    const float u_step = n_new > 1 ? (u_end - u_start) / (float)(n_new - 1) : 0.f;   // linspace(steps=1) = [start]
    int ptr = 0;   // number of cdf entries <= u (searchsorted right=True); cdf and u are both non-decreasing
    for (int jj = 0; jj < n_new; ++jj) {
        const float u = (jj < n_new / 2) ? u_start + (float)jj * u_step : u_end - (float)(n_new - 1 - jj) * u_step;
        for (;;) {   // four entries per round trip (the cdf is non-decreasing: the entries <= u are a prefix)
            int adv = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) adv += (ptr + q < k && cr[ptr + q < k ? ptr + q : k - 1] <= u) ? 1 : 0;
            ptr += adv;
            if (adv < 4) break;
        }
        const int below = ptr - 1 > 0 ? ptr - 1 : 0;
        const int above = ptr < k - 1 ? ptr : k - 1;
        const float c_lo = cr[below], c_hi = cr[above];
        const float b_lo = zr[below], b_hi = zr[above];
        float denom = c_hi - c_lo;
        denom = denom < 1e-5f ? 1.f : denom;
        const float t = (u - c_lo) / denom;
        z_new[(size_t)ray * n_new + jj] = b_lo + t * (b_hi - b_lo);
        if (inds_out != nullptr) inds_out[(size_t)ray * n_new + jj] = ptr;
    }
}

// One WAVE per ray, for the small batches of the fitting loops (196 rays: thread-per-ray leaves 252 of 256 CUs idle and
// walks ~100 dependent steps of two exps each, ~50 us).  The element-wise part (slopes, the two sigmoids, alpha) is
// computed by all lanes; the two prefix recurrences stay strictly sequential in lane 0, in the same order and with the
// same operations as above (torch.cumprod / torch.cumsum order: the sample indices must not change), and the 16
// inversions run one per lane.  Same results bit for bit as k_upsample_tiled / k_upsample_direct.
// UpsExtra (the two-field render's importance rounds, where this launch sat in front of a column copy and a sample-point launch in a
// chain of dependent launches): the new depths also go to columns col .. col + n_new - 1 of the [n_rays, ld] list `zcat`, and, with
// `pts`, the new sample positions o + d z (k_sample_points' statement, mid = 0) are written as well.
// Pre-merge (the same rounds: cat_z_vals of the PREVIOUS round, utils/renderer.py:88-105, sat between that round's sdf launch and
// this launch): with m_prev > 0 the row is the stable merge of z / sdf [k - m_prev] with zp / sp [m_prev] -- k_merge's ranks: old
// sample a_i at i + #{j: b_j < a_i}, new sample b_j at j + #{i: a_i <= b_j}, the sdf rows taken from ray % quirk_p when quirk_p > 0
// (SURVEY B-1) -- assembled in LDS, used from there, and written to z_out / sdf_out [n_rays, k] for the rounds after this one.
// Gather (the hand's coarse round under the far-field skip): with `pos` the sdf row is read through the compaction record --
// sdf_c[pos[i]], the far sample's value sdf_c[n_dev - 1] where pos[i] < 0 (k_hand_scatter_sdf's statement) -- and the dense row
// goes to sdf_out.
struct UpsExtra {
    float* zcat;
    int ld, col;
    const float *o, *d;
    float* pts;
    const float *zp, *sp;
    int m_prev, quirk_p;
    float *z_out, *sdf_out;
    const int *pos, *n_dev;
    const float* sdf_c;
};
__global__ __launch_bounds__(256) void k_upsample_wave(const float* __restrict__ z, const float* __restrict__ sdf, int n_rays, int k,
                                                       int n_new, float inv_s, float* __restrict__ z_new,
                                                       int64_t* __restrict__ inds_out, UpsExtra ex) {
    __shared__ float lz[4][UPS_MAX_K], ls[4][UPS_MAX_K], lw[4][UPS_MAX_K], lb[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ray = blockIdx.x * 4 + wv;
    if (ray >= n_rays) return;
    float* zr = lz[wv];
    float* sr = ls[wv];
    float* wr = lw[wv];
    if (ex.m_prev > 0) {
        // the previous round's cat_z_vals: a = the old row (staged in wr, free until the weights are formed), b = that round's depths
        const int m = ex.m_prev, k0 = k - m;
        const int srow = ex.quirk_p > 0 ? ray % ex.quirk_p : ray;
        float* br = lb[wv];
        constexpr int RMAX = UPS_MAX_K / 64;
        float as[RMAX];
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            const int i = r * 64 + lane;
            const bool ok = i < k0;
            if (ok) wr[i] = z[(size_t)ray * k0 + i];
            as[r] = ok ? sdf[(size_t)srow * k0 + i] : 0.f;
        }
        float bv = 0.f, bs = 0.f;
        if (lane < m) {
            bv = ex.zp[(size_t)ray * m + lane];
            bs = ex.sp[(size_t)srow * m + lane];
            br[lane] = bv;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            const int i = r * 64 + lane;
            if (i < k0) {
                const float a = wr[i];
                int lo = 0, hi = m;   // #{j: b_j < a} (b sorted)
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (br[mid] < a)
                        lo = mid + 1;
                    else
                        hi = mid;
                }
                zr[i + lo] = a;
                sr[i + lo] = as[r];
            }
        }
        if (lane < m) {
            int lo = 0, hi = k0;   // #{i: a_i <= b} (a sorted)
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (wr[mid] <= bv)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            zr[lane + lo] = bv;
            sr[lane + lo] = bs;
        }
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < k; i += 64) {
            ex.z_out[(size_t)ray * k + i] = zr[i];
            ex.sdf_out[(size_t)ray * k + i] = sr[i];
        }
    } else if (ex.pos != nullptr) {
        const int far = ex.n_dev[0] - 1;
        for (int i = lane; i < k; i += 64) {
            const size_t g = (size_t)ray * k + i;
            const int q = ex.pos[g];
            const float v = ex.sdf_c[q >= 0 ? q : far];
            zr[i] = z[g];
            sr[i] = v;
            ex.sdf_out[g] = v;
        }
    } else {
        for (int i = lane; i < k; i += 64) {
            zr[i] = z[(size_t)ray * k + i];
            sr[i] = sdf[(size_t)ray * k + i];
        }
    }
    __builtin_amdgcn_wave_barrier();
    // alpha of section i (element-wise in the reference too: utils/renderer.py:68-81)
    for (int i = lane; i + 1 < k; i += 64) {
        const float z0 = zr[i], z1 = zr[i + 1], s0 = sr[i], s1 = sr[i + 1];
        const float mid_sdf = (s0 + s1) * 0.5f;
        const float cosv = (s1 - s0) / (z1 - z0 + 1e-5f);
        const float prev_cos = i > 0 ? (s0 - sr[i - 1]) / (z0 - zr[i - 1] + 1e-5f) : 0.f;
        float c = fminf(prev_cos, cosv);
        c = fminf(fmaxf(c, -1e3f), 0.f);
        const float dist = z1 - z0;
        const float prev_cdf = sigmoid_acc((mid_sdf - c * dist * 0.5f) * inv_s);
        const float next_cdf = sigmoid_acc((mid_sdf + c * dist * 0.5f) * inv_s);
        wr[i + 1] = (prev_cdf - next_cdf + 1e-5f) / (prev_cdf + 1e-5f);
    }
    __builtin_amdgcn_wave_barrier();
    // The two prefix recurrences stay strictly sequential, in torch.cumprod / torch.cumsum order and with the statements of
    // UpsState::step -- but they run on REGISTERS, every lane computing the same (uniform) chain with element e fetched by
    // v_readlane from the lane that holds it (slice r = e / 64, lane e % 64).  As loops of lane 0 over the LDS rows each of the
    // ~2 k steps paid an LDS round trip inside its dependency chain: ~10 us of a ~20 us launch that sits four times on the critical
    // path of a fitting step.
    constexpr int UPS_SL = UPS_MAX_K / 64;
    float areg[UPS_SL], wreg[UPS_SL];
#pragma unroll
    for (int r = 0; r < UPS_SL; ++r) {
        const int e = r * 64 + lane;
        areg[r] = (e >= 1 && e < k) ? wr[e] : 0.f;   // alpha of section e - 1
        wreg[r] = 0.f;
    }
    float sum = 0.f;
    {
        float T = 1.f;
#pragma unroll
        for (int r = 0; r < UPS_SL; ++r) {
            const int e_hi = k - r * 64 < 64 ? k - r * 64 : 64;
            for (int l = (r == 0 ? 1 : 0); l < e_hi; ++l) {
                const float alpha = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, areg[r]), l));
                const float w = alpha * T + 1e-5f;
                T = T * (1.f - alpha + 1e-7f);
                sum += w;
                wreg[r] = lane == l ? w : wreg[r];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < UPS_SL; ++r) wreg[r] = wreg[r] / sum;   // pdf (element-wise)
    {   // cdf = [0, cumsum(pdf)], sequential
        float run = 0.f;
        float creg[UPS_SL];
#pragma unroll
        for (int r = 0; r < UPS_SL; ++r) {
            creg[r] = 0.f;
            const int e_hi = k - r * 64 < 64 ? k - r * 64 : 64;
            for (int l = (r == 0 ? 1 : 0); l < e_hi; ++l) {
                run += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wreg[r]), l));
                creg[r] = lane == l ? run : creg[r];
            }
        }
#pragma unroll
        for (int r = 0; r < UPS_SL; ++r) {
            const int e = r * 64 + lane;
            if (e < k) wr[e] = creg[r];   // (entry 0 is 0)
        }
    }
    __builtin_amdgcn_wave_barrier();
    const float u_start = (float)(0.0 + 0.5 / (double)n_new), u_end = (float)(1.0 - 0.5 / (double)n_new);   // (torch.linspace of Python doubles: one rounding each)
    const float u_step = n_new > 1 ? (u_end - u_start) / (float)(n_new - 1) : 0.f;
    for (int jj = lane; jj < n_new; jj += 64) {
        const float u = (jj < n_new / 2) ? u_start + (float)jj * u_step : u_end - (float)(n_new - 1 - jj) * u_step;
        // searchsorted(right=True) on the non-decreasing cdf: the number of entries <= u, by bisection (a scan from the left paid
        // an LDS round trip per entry)
        int ptr = 0;
        {
            int hi = k;
            while (ptr < hi) {
                const int mid = (ptr + hi) >> 1;
                if (wr[mid] <= u)
                    ptr = mid + 1;
                else
                    hi = mid;
            }
        }
        const int below = ptr - 1 > 0 ? ptr - 1 : 0;
        const int above = ptr < k - 1 ? ptr : k - 1;
        const float c_lo = wr[below], c_hi = wr[above];
        const float b_lo = zr[below], b_hi = zr[above];
        float denom = c_hi - c_lo;
        denom = denom < 1e-5f ? 1.f : denom;
        const float t = (u - c_lo) / denom;
        const float zn = b_lo + t * (b_hi - b_lo);
        z_new[(size_t)ray * n_new + jj] = zn;
        if (inds_out != nullptr) inds_out[(size_t)ray * n_new + jj] = ptr;
        if (ex.zcat != nullptr) ex.zcat[(size_t)ray * ex.ld + ex.col + jj] = zn;
        if (ex.pts != nullptr) {
            const size_t q = (size_t)ray * n_new + jj;
#pragma unroll
            for (int c = 0; c < 3; ++c) ex.pts[3 * q + c] = ex.o[3 * ray + c] + ex.d[3 * ray + c] * zn;
        }
    }
}

// ---- cat_z_vals (utils/renderer.py:88-105): stable merge of two sorted rows ----------------------
__global__ void k_merge_serial(const float* __restrict__ z, const float* __restrict__ z_new, const float* __restrict__ sdf,
                        const float* __restrict__ sdf_new, int n_rays, int k, int m, int quirk_p,
                        float* __restrict__ z_out, float* __restrict__ sdf_out, int64_t* __restrict__ index) {
    const int ray = blockIdx.x * blockDim.x + threadIdx.x;
    if (ray >= n_rays) return;
    const float* a = z + (size_t)ray * k;
    const float* b = z_new + (size_t)ray * m;
    // SURVEY B-1: the batched renderer gathers SDF values from frame 0's row of the same pixel
    const int srow = quirk_p > 0 ? ray % quirk_p : ray;
    const float* sa = sdf ? sdf + (size_t)srow * k : nullptr;
    const float* sb = sdf_new ? sdf_new + (size_t)srow * m : nullptr;
    float* zo = z_out + (size_t)ray * (k + m);
    int i = 0, jj = 0;
    for (int o = 0; o < k + m; ++o) {
        bool take_a;
        if (i >= k) take_a = false;
        else if (jj >= m) take_a = true;
        else take_a = a[i] <= b[jj];          // ties: the old sample first (stable sort of cat([z, z_new]))
        if (take_a) {
            zo[o] = a[i];
            if (sdf_out) sdf_out[(size_t)ray * (k + m) + o] = sa[i];
            if (index) index[(size_t)ray * (k + m) + o] = i;
            ++i;
        } else {
            zo[o] = b[jj];
            if (sdf_out) sdf_out[(size_t)ray * (k + m) + o] = sb[jj];
            if (index) index[(size_t)ray * (k + m) + o] = k + jj;
            ++jj;
        }
    }
}

// The same merge by ranks, one wave per ray (k <= 256, m <= 64): element a_i lands at i + #{j: b_j < a_i}, element
// b_j at j + #{i: a_i <= b_j} (ties: the old sample first, as the stable sort of cat([z, z_new]) places them).
// Rows are read and written with consecutive lanes on consecutive addresses; the serial kernel above walks one
// row per lane (row stride k floats: every access its own cache line) and runs at 4 % of the HBM rate.
template <int R>   // R = ceil(k / 64): 64-element slices of the old row per lane (a k = 64 row needs one, not four)
struct MergeRow {
    float av[R], as[R], bv, bs;
    __device__ __forceinline__ void load(const float* __restrict__ z, const float* __restrict__ z_new, const float* __restrict__ sdf,
                                         const float* __restrict__ sdf_new, int ray, int srow, int k, int m, int lane, bool with_sdf) {
        const float* a = z + (size_t)ray * k;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int e = r * 64 + lane;
            const bool ok = e < k;
            av[r] = ok ? a[e] : 0.f;
            as[r] = (ok && with_sdf) ? sdf[(size_t)srow * k + e] : 0.f;
        }
        bv = lane < m ? z_new[(size_t)ray * m + lane] : 0.f;
        bs = (lane < m && with_sdf) ? sdf_new[(size_t)srow * m + lane] : 0.f;
    }
    // Ranks by two binary searches over lane-resident sorted rows (ds_bpermute): cnt_i = #{j: b_j < a_i} (search b, <= 7 steps), and
    // since b is sorted {j: b_j < a_i} = [0, cnt_i), so #{i: a_i <= b_j} = #{i: cnt_i <= j} (search the non-decreasing cnt, <= 9 steps).
    // The m-step counting loop this replaces (readlane + 2 compares + ballot per new depth) was the kernel's time: ~130 of ~170
    // instructions per ray, VALU-issue bound at 0.46 of the HBM peak.
    __device__ __forceinline__ void rank_to_lds(int k, int m, int lane, float* st_z, float* st_s, int* st_i) const {
        int cnt[R];
        const int s_b = 1 << (31 - __builtin_clz(m)), s_a = 1 << (31 - __builtin_clz(k));   // (uniform) largest powers of two <= m, k
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int pos = 0;
            for (int st = s_b; st >= 1; st >>= 1) {
                const int idx = pos + st;
                const float v = __shfl(bv, idx - 1, 64);
                pos = (idx <= m && v < av[r]) ? idx : pos;
            }
            cnt[r] = pos;
        }
        int bcnt = 0;
        for (int st = s_a; st >= 1; st >>= 1) {
            const int idx = bcnt + st;
            const int e = idx - 1;
            int c = __shfl(cnt[0], e & 63, 64);
#pragma unroll
            for (int r = 1; r < R; ++r) {
                const int cr = __shfl(cnt[r], e & 63, 64);
                c = (e >> 6) == r ? cr : c;
            }
            bcnt = (idx <= k && c <= lane) ? idx : bcnt;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int e = r * 64 + lane;
            if (e < k) {
                const int o = e + cnt[r];
                st_z[o] = av[r];
                st_s[o] = as[r];
                st_i[o] = e;
            }
        }
        if (lane < m) {
            const int o = lane + bcnt;
            st_z[o] = bv;
            st_s[o] = bs;
            st_i[o] = k + lane;
        }
    }
};
// Two ADJACENT rays per wave and iteration (four: measured slower): both rows are requested before either is ranked (a wave moves ~2 KB per ray with a
// dependent load -> rank -> store chain in between), the merged rows are assembled in a wave-private LDS region and
// written as ONE contiguous block with consecutive lanes on consecutive addresses (scattered by rank, every store instruction
// touched each cache line of a row and filled part of it).  0.46 -> 0.6 of the HBM peak at k = 64.
template <int R>
__global__ __launch_bounds__(256) void k_merge(const float* __restrict__ z, const float* __restrict__ z_new,
                                               const float* __restrict__ sdf, const float* __restrict__ sdf_new, int n_rays,
                                               int k, int m, int quirk_p, float* __restrict__ z_out,
                                               float* __restrict__ sdf_out, int64_t* __restrict__ index) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int stride = gridDim.x * 8;
    const bool with_sdf = sdf_out != nullptr;
    constexpr int ROWS2 = 2 * (64 * R + 64);   // 2 rows of k + m <= 64 R + 64 entries
    __shared__ float st_zs[4][2][ROWS2];   // [wave][z | sdf][..]
    __shared__ int st_is[4][ROWS2];
    float *st_z = st_zs[wave][0], *st_s = st_zs[wave][1];
    int* st_i = st_is[wave];
    const int w = k + m;
    for (int ray = blockIdx.x * 8 + 2 * wave; ray < n_rays; ray += stride) {
        const int ray2 = ray + 1;
        const bool two = ray2 < n_rays;
        MergeRow<R> r0, r1;
        r0.load(z, z_new, sdf, sdf_new, ray, quirk_p > 0 ? ray % quirk_p : ray, k, m, lane, with_sdf);   // (SURVEY B-1: frame 0's sdf row)
        if (two) r1.load(z, z_new, sdf, sdf_new, ray2, quirk_p > 0 ? ray2 % quirk_p : ray2, k, m, lane, with_sdf);
        r0.rank_to_lds(k, m, lane, st_z, st_s, st_i);
        if (two) r1.rank_to_lds(k, m, lane, st_z + w, st_s + w, st_i + w);
        __builtin_amdgcn_wave_barrier();   // (a wave's LDS operations complete in order; this only pins the compiler's order)
        const size_t ob = (size_t)ray * w;
        const int total = two ? 2 * w : w;
        for (int p = lane; p < total; p += 64) {
            z_out[ob + p] = st_z[p];
            if (sdf_out) sdf_out[ob + p] = st_s[p];
            if (index) index[ob + p] = st_i[p];
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- row sort (utils/renderer.py:498): rank sort, one wave per row -------------------------------
constexpr int SORT_MAX_N = 1024;
__global__ __launch_bounds__(64) void k_sort_rows(const float* __restrict__ v, int n_rows, int n, float* __restrict__ out) {
    __shared__ float row[SORT_MAX_N];
    const int r = blockIdx.x;
    const int lane = threadIdx.x;
    for (int i = lane; i < n; i += 64) row[i] = v[(size_t)r * n + i];
    __syncthreads();
    for (int i = lane; i < n; i += 64) {
        const float x = row[i];
        int rank = 0;
        for (int jj = 0; jj < n; ++jj) {
            const float y = row[jj];
            rank += (y < x || (y == x && jj < i)) ? 1 : 0;
        }
        out[(size_t)r * n + rank] = x;
    }
}

// ---- host launchers ----------------------------------------------------------------------------
static inline dim3 grid1d(size_t n, int block) { return dim3((unsigned)((n + block - 1) / block)); }

int ray_gen(const float* xy, const float* R, const float* T, const float* focal, const float* principal, int n_cams,
            int rays_per_cam, float* rays_o, float* rays_d, hipStream_t s) {
    HN_REQUIRE(n_cams > 0 && rays_per_cam >= 0, "bad ray_gen sizes");
    const size_t n = (size_t)n_cams * rays_per_cam;
    if (n == 0) return HN_OK;
    hipLaunchKernelGGL(k_ray_gen, grid1d(n, 256), dim3(256), 0, s, xy, R, T, focal, principal, n_cams, rays_per_cam,
                       rays_o, rays_d);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

int obj_local_fwd(const float* o, const float* d, const float* Ro, const float* To, int n_frames, int rpf, float* oo,
                  float* dd, hipStream_t s, bool transposed) {
    HN_REQUIRE(n_frames > 0 && rpf >= 0, "bad obj_local sizes");
    const size_t n = (size_t)n_frames * rpf;
    if (n == 0) return HN_OK;
    hipLaunchKernelGGL(k_obj_local_fwd, grid1d(n, 256), dim3(256), 0, s, o, d, Ro, To, (int)n, rpf, oo, dd, transposed ? 1 : 0);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

int obj_local_bwd(const float* o, const float* d, const float* Ro, const float* To, const float* go, const float* gd,
                  int n_frames, int rpf, float* g_o, float* g_d, float* g_Ro, float* g_To, hipStream_t s) {
    HN_REQUIRE(n_frames > 0 && rpf >= 0, "bad obj_local sizes");
    hipLaunchKernelGGL(k_obj_local_bwd, dim3(n_frames), dim3(256), 0, s, o, d, Ro, To, go, gd, rpf, g_o, g_d, g_Ro,
                       g_To);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

int coarse_z(const float* t_rand, int n_rays, int n, float near, float span, float sample_dist, float* z,
             hipStream_t s) {
    HN_REQUIRE(n >= 2, "n_samples must be >= 2");
    if (n_rays == 0) return HN_OK;
    hipLaunchKernelGGL(k_coarse_z, grid1d((size_t)n_rays * n, 256), dim3(256), 0, s, t_rand, n_rays, n, near, span,
                       sample_dist, z);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

int sample_points(const float* o, const float* d, const float* z, int n_rays, int n, int mid, float sample_dist,
                  float* pts, float* dists, hipStream_t s) {
    HN_REQUIRE(!mid || dists != nullptr, "dists required for mid-point sampling");
    if (n_rays == 0 || n == 0) return HN_OK;
    const bool aligned = ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(pts) | reinterpret_cast<uintptr_t>(dists)) & 15) == 0;
    if (n % 4 == 0 && aligned && (size_t)n_rays * n < (1u << 31))
        hipLaunchKernelGGL(k_sample_points_t, grid1d((size_t)n_rays * n / 4, 256), dim3(256), 0, s, o, d, z, n_rays, n, mid,
                           sample_dist, pts, dists);
    else
        hipLaunchKernelGGL(k_sample_points, grid1d((size_t)n_rays * n, 256), dim3(256), 0, s, o, d, z, n_rays, n, mid,
                           sample_dist, pts, dists);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

int sample_points_bwd(const float* z, const float* g_pts, int n_rays, int n, int mid, float sample_dist, float* g_o,
                      float* g_d, hipStream_t s) {
    if (n_rays == 0) return HN_OK;
    HN_REQUIRE(n >= 1, "n must be positive");
    hipLaunchKernelGGL(k_sample_points_bwd, dim3((n_rays + 3) / 4), dim3(256), 0, s, z, g_pts, n_rays, n, mid, sample_dist,
                       g_o, g_d);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

int dual_prologue(const float* o, const float* d, const float* Ro, const float* To, int n_frames, int rpf, float* o_out, float* d_out,
                  const float* t_rand, int n, float near, float span, float sample_dist, float* z_hand, float* z_obj, float* zcat, int S,
                  hipStream_t s, bool transposed) {
    HN_REQUIRE(n >= 2 && n_frames > 0 && rpf >= 0, "bad prologue sizes");
    const size_t n_rays = (size_t)n_frames * rpf;
    if (n_rays == 0) return HN_OK;
    hipLaunchKernelGGL(k_dual_prologue, grid1d(n_rays * n, 256), dim3(256), 0, s, o, d, Ro, To, (int)n_rays, rpf, o_out, d_out, t_rand, n, near, span,
                       sample_dist, z_hand, z_obj, zcat, S, transposed ? 1 : 0);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

int obj_rays_bwd(const float* z, const float* g_pts, int n_frames, int rpf, int n, float sample_dist, const float* gd_alpha, const float* gd_colour,
                 const float* o, const float* d, const float* Ro, const float* To, float* g_o, float* g_d, float* g_Ro, float* g_To, hipStream_t s,
                 bool transposed, float* part, unsigned* counter, const float* gd_alpha_samples, const float* gd_colour_samples) {
    const int n_rays = n_frames * rpf;
    if (n_rays == 0) return HN_OK;
    HN_REQUIRE(n >= 1 && rpf >= 1 && g_Ro != nullptr && g_To != nullptr && (gd_alpha != nullptr || gd_alpha_samples != nullptr), "obj_rays_bwd: bad arguments");
    float* const rows = counter != nullptr ? part : nullptr;
    const bool own_launch = rows != nullptr && n_frames >= 3;   // (the rows' sums per frame: in the kernel's last block, or a block per frame behind it)
    hipLaunchKernelGGL(k_obj_rays_bwd, dim3((n_rays + 3) / 4), dim3(256), 0, s, z, g_pts, n_rays, n, sample_dist, gd_alpha, gd_colour, o, d, Ro, To, rpf,
                       g_o, g_d, g_Ro, g_To, transposed ? 1 : 0, rows, counter, gd_alpha_samples, gd_colour_samples, own_launch ? 0 : 1);
    if (own_launch) hipLaunchKernelGGL(k_obj_rays_finalize, dim3(n_frames), dim3(256), 0, s, rows, n_rays, rpf, g_Ro, g_To);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

int upsample(const float* z, const float* sdf, int n_rays, int k, int n_new, float inv_s, float* z_new, int64_t* inds,
             hipStream_t s) {
    HN_REQUIRE(k >= 2 && k <= UPS_DIRECT_MAX_K && n_new >= 1, "upsample: k=%d n_new=%d out of range (2 <= k <= %d)", k, n_new, UPS_DIRECT_MAX_K);
    if (n_rays == 0) return HN_OK;
    if (k > UPS_MAX_K || n_new > 64) {
        // beyond the shapes of the confs (k <= 112, 16 new depths per round): the thread-per-ray form, whose only limit is its cdf row in LDS
        // -- the same operations in the same order as the forms below
        const size_t lds = (size_t)k * 64 * sizeof(float);
        static std::atomic<uint64_t> lds_direct{0};
        if (lds > 64 * 1024) HN_TRY_RC(ensure_dynamic_lds(reinterpret_cast<const void*>(k_upsample_direct), (int)((size_t)UPS_DIRECT_MAX_K * 64 * sizeof(float)), &lds_direct));
        hipLaunchKernelGGL(k_upsample_direct, grid1d(n_rays, 64), dim3(64), lds, s, z, sdf, n_rays, k, n_new, inv_s, z_new, inds);
        HN_LAUNCH_CHECK();
        return HN_OK;
    }
    if (n_rays <= 8192) {   // small batches (the fitting loops): one wave per ray
        hipLaunchKernelGGL(k_upsample_wave, dim3((n_rays + 3) / 4), dim3(256), 0, s, z, sdf, n_rays, k, n_new, inv_s, z_new, inds,
                           UpsExtra{nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr});
        HN_LAUNCH_CHECK();
        return HN_OK;
    }
    if (k <= 224) {   // (64 x (k + 19) floats of LDS per wave: within the 64 KiB a launch gets without an attribute)
        hipLaunchKernelGGL(k_upsample_tiled, grid1d(n_rays, 64), dim3(64), (size_t)64 * (k + 1 + 18) * sizeof(float), s, z, sdf, n_rays,
                           k, n_new, inv_s, z_new, inds);
    } else {
        hipLaunchKernelGGL(k_upsample_direct, grid1d(n_rays, 64), dim3(64), (size_t)k * 64 * sizeof(float), s, z, sdf, n_rays,
                           k, n_new, inv_s, z_new, inds);
    }
    HN_LAUNCH_CHECK();
    return HN_OK;
}

// up_sample of a small batch (<= 8192 rays: the fitting loops) with the round's follow-up writes in the same launch (UpsExtra);
// HN_EINVAL-free fallback for larger batches: false is returned and the caller runs the three launches
bool upsample_fused_ok(int n_rays, int k, int n_new) { return n_rays > 0 && n_rays <= 8192 && k >= 2 && k <= UPS_MAX_K && n_new >= 1 && n_new <= 64; }
bool upsample_fused(const float* z, const float* sdf, int n_rays, int k, int n_new, float inv_s, float* z_new, float* zcat, int ld, int col,
                    const float* o, const float* d, float* pts, hipStream_t s, const UpsPre* pre) {
    if (!upsample_fused_ok(n_rays, k, n_new)) return false;
    UpsExtra ex{zcat, ld, col, o, d, pts, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (pre != nullptr) {
        if (pre->m_prev > 0) {
            if (pre->m_prev > 64 || pre->m_prev >= k || pre->zp == nullptr || pre->sp == nullptr || pre->z_out == nullptr || pre->sdf_out == nullptr)
                return false;
            ex.zp = pre->zp, ex.sp = pre->sp, ex.m_prev = pre->m_prev, ex.quirk_p = pre->quirk_p, ex.z_out = pre->z_out, ex.sdf_out = pre->sdf_out;
        } else if (pre->pos != nullptr) {
            if (pre->n_dev == nullptr || pre->sdf_c == nullptr || pre->sdf_out == nullptr) return false;
            ex.pos = pre->pos, ex.n_dev = pre->n_dev, ex.sdf_c = pre->sdf_c, ex.sdf_out = pre->sdf_out;
        }
    }
    hipLaunchKernelGGL(k_upsample_wave, dim3((n_rays + 3) / 4), dim3(256), 0, s, z, sdf, n_rays, k, n_new, inv_s, z_new, (int64_t*)nullptr, ex);
    return hipGetLastError() == hipSuccess;
}

int merge(const float* z, const float* z_new, const float* sdf, const float* sdf_new, int n_rays, int k, int m,
          int quirk_p, float* z_out, float* sdf_out, int64_t* index, hipStream_t s) {
    HN_REQUIRE((sdf_out == nullptr) || (sdf != nullptr && sdf_new != nullptr), "merge: sdf inputs missing");
    if (n_rays == 0) return HN_OK;
    if (k <= 256 && m <= 64) {
        const int slices = (k + 63) / 64;
        const int blocks = (n_rays + 7) / 8 < 8192 ? (n_rays + 7) / 8 : 8192;   // 4 waves x 2 adjacent rays per block and iteration
        if (slices <= 1)
            hipLaunchKernelGGL(k_merge<1>, dim3(blocks), dim3(256), 0, s, z, z_new, sdf, sdf_new, n_rays, k, m, quirk_p, z_out, sdf_out, index);
        else if (slices == 2)
            hipLaunchKernelGGL(k_merge<2>, dim3(blocks), dim3(256), 0, s, z, z_new, sdf, sdf_new, n_rays, k, m, quirk_p, z_out, sdf_out, index);
        else
            hipLaunchKernelGGL(k_merge<4>, dim3(blocks), dim3(256), 0, s, z, z_new, sdf, sdf_new, n_rays, k, m, quirk_p, z_out, sdf_out, index);
    } else {
        hipLaunchKernelGGL(k_merge_serial, grid1d(n_rays, 64), dim3(64), 0, s, z, z_new, sdf, sdf_new, n_rays, k, m, quirk_p,
                           z_out, sdf_out, index);
    }
    HN_LAUNCH_CHECK();
    return HN_OK;
}

int sort_rows(const float* v, int n_rows, int n, float* out, hipStream_t s) {
    HN_REQUIRE(n >= 1 && n <= SORT_MAX_N, "sort_rows: n=%d out of range", n);
    if (n_rows == 0) return HN_OK;
    hipLaunchKernelGGL(k_sort_rows, dim3(n_rows), dim3(64), 0, s, v, n_rows, n, out);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

}  // namespace hn
