// The non-skeleton part of the fitting loops' pose side as launches instead of ~200 small torch operators per step:
//   hn_rigid_pose   palm rigid motion of the predicted joints and its inverse folded into the bone transformations
//                   (fitting_single.py:213-217), object refinement obj_r = rot6d(obj_rot_refine) ori_obj_r,
//                   obj_t = ori_obj_t + obj_trans_refine (:227-231), joint loss pose_loss(joint3d_pred, joint_3d) (:119-122,
//                   :260) -- values and the exact Jacobian (forward-mode dual numbers, one thread per input), like
//                   hn_pose_chain
//   hn_verts_loss   pose_loss over the object's vertices (:232-233: mean distance between the vertex sets of two rigid
//                   poses) with its closed-form gradient w.r.t. the first pose
//   hn_jacobian_vjp g_in = J^T g_out for a stored Jacobian (the backward pass of the two dual-number ops)
#include <math.h>

#include "hn_common.h"
#include "hn_pose_chain.h"

namespace hn {

using pose::Dual;
using pose::M3;
using pose::V3;
constexpr int RIGID_IN = 18;    // [obj_rot6 | obj_trans3 | palm_rot6 | palm_trans3]
constexpr int RIGID_OUT = 412;  // [bt_inv 336 | joint_3d 63 | obj_r 9 | obj_t 3 | joint_loss 1]

// with_palm = 0: the object half only (outputs 399 .. 410; the others are left untouched) -- the hand half then comes from
// hn_pose_chain
__global__ __launch_bounds__(32) void k_rigid_pose(const float* __restrict__ bt_inv0, const float* __restrict__ joints0, const float* __restrict__ Ro_pred,
                                                   const float* __restrict__ To_pred, const float* __restrict__ params, int n_frames, int with_palm,
                                                   float* __restrict__ out, float* __restrict__ jac) {
    using T = Dual<double>;
    const int f = blockIdx.x, k = threadIdx.x;   // k = 0: values; 1 + input: that input's derivative
    if (f >= n_frames || k > RIGID_IN || (k > 0 && jac == nullptr)) return;
    T x[RIGID_IN];
    for (int i = 0; i < RIGID_IN; ++i) x[i] = T((double)params[(size_t)f * RIGID_IN + i], i == k - 1 ? 1.0 : 0.0);
    float* o = out + (size_t)f * RIGID_OUT;
    float* J = jac != nullptr ? jac + (size_t)f * RIGID_OUT * RIGID_IN : nullptr;
    auto put = [&](int idx, const T& v) {
        if (k == 0)
            o[idx] = (float)v.v;
        else
            J[idx * RIGID_IN + (k - 1)] = (float)v.d;
    };
    {   // object: obj_r = rot6d(obj_rot) @ Ro_pred, obj_t = To_pred + obj_trans
        const T r6[6] = {x[0], x[1], x[2], x[3], x[4], x[5]};
        const M3<T> R = pose::rot6d_to_matrix<double>(r6);
        M3<T> P;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) P.m[i][j] = T((double)Ro_pred[(size_t)f * 9 + 3 * i + j]);
        const M3<T> Ro = pose::mul(R, P);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) put(399 + 3 * i + j, Ro.m[i][j]);
        for (int c = 0; c < 3; ++c) put(408 + c, x[6 + c] + (double)To_pred[(size_t)f * 3 + c]);
    }
    if (!with_palm) return;
    const T r6[6] = {x[9], x[10], x[11], x[12], x[13], x[14]};
    const M3<T> Rp = pose::rot6d_to_matrix<double>(r6);
    const V3<T> tp = {{x[15], x[16], x[17]}};
    const float* j0 = joints0 + (size_t)f * 63;
    const V3<T> root = {{T((double)j0[0]), T((double)j0[1]), T((double)j0[2])}};
    // joint_3d = R_palm (joints - root) + root + T_palm ; joint loss = sum_j |joints_j - joint_3d_j| / 21
    T jl = T(0.0);
    for (int j = 0; j < 21; ++j) {
        const V3<T> p = {{T((double)j0[3 * j]), T((double)j0[3 * j + 1]), T((double)j0[3 * j + 2])}};
        const V3<T> q = pose::mul(Rp, p - root) + root + tp;
        for (int c = 0; c < 3; ++c) put(336 + 3 * j + c, q.x[c]);
        jl = jl + pose::norm(p - q);
    }
    put(411, jl * (1.0 / 21.0));
    // G p = R_palm (p - root) + root + T  ->  G^-1 = (R^T | root - R^T (root + T));  bt_inv = bt_inv0 G^-1
    const M3<T> Rt = pose::transpose(Rp);
    const V3<T> ti = root - pose::mul(Rt, root + tp);
    for (int b = 0; b < 21; ++b) {
        const float* m = bt_inv0 + ((size_t)f * 21 + b) * 16;
        for (int r = 0; r < 4; ++r) {
            const T m0 = T((double)m[4 * r]), m1 = T((double)m[4 * r + 1]), m2 = T((double)m[4 * r + 2]), m3 = T((double)m[4 * r + 3]);
            for (int c = 0; c < 3; ++c) put(16 * b + 4 * r + c, m0 * Rt.m[0][c] + m1 * Rt.m[1][c] + m2 * Rt.m[2][c]);
            put(16 * b + 4 * r + 3, m0 * ti.x[0] + m1 * ti.x[1] + m2 * ti.x[2] + m3);
        }
    }
}

// loss[p] = (1 / V) sum_v |(Ra - Rb) v + (ta - tb)|;  gR[p] = d loss / d Ra = (1 / V) sum_v u_v v^T,  gt[p] = (1 / V) sum_v u_v
// with u_v the unit residual (0 where the residual is 0: torch.norm's convention).  One 256-thread block per pair.
__global__ __launch_bounds__(256) void k_verts_loss(const float* __restrict__ Ra, const float* __restrict__ ta, const float* __restrict__ Rb,
                                                    const float* __restrict__ tb, const float* __restrict__ verts, int n_verts, float* __restrict__ loss,
                                                    float* __restrict__ gR, float* __restrict__ gt) {
    const int p = blockIdx.x;
    float D[9], d[3];
    for (int i = 0; i < 9; ++i) D[i] = Ra[p * 9 + i] - Rb[p * 9 + i];
    for (int i = 0; i < 3; ++i) d[i] = ta[p * 3 + i] - tb[p * 3 + i];
    float acc[13];
    for (int i = 0; i < 13; ++i) acc[i] = 0.f;
    for (int v = threadIdx.x; v < n_verts; v += blockDim.x) {
        const float x = verts[3 * v], y = verts[3 * v + 1], z = verts[3 * v + 2];
        const float e0 = D[0] * x + D[1] * y + D[2] * z + d[0], e1 = D[3] * x + D[4] * y + D[5] * z + d[1], e2 = D[6] * x + D[7] * y + D[8] * z + d[2];
        const float n = sqrtf(e0 * e0 + e1 * e1 + e2 * e2);
        const float inv = n > 0.f ? 1.f / n : 0.f;
        const float u0 = e0 * inv, u1 = e1 * inv, u2 = e2 * inv;
        acc[0] += n;
        acc[1] += u0 * x; acc[2] += u0 * y; acc[3] += u0 * z;
        acc[4] += u1 * x; acc[5] += u1 * y; acc[6] += u1 * z;
        acc[7] += u2 * x; acc[8] += u2 * y; acc[9] += u2 * z;
        acc[10] += u0; acc[11] += u1; acc[12] += u2;
    }
    __shared__ float red[13][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = 0; i < 13; ++i) {
        float s = acc[i];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) red[i][wave] = s;
    }
    __syncthreads();
    if (threadIdx.x < 13) {
        const float s = (red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]) / (float)n_verts;
        if (threadIdx.x == 0)
            loss[p] = s;
        else if (threadIdx.x < 10)
            gR[p * 9 + threadIdx.x - 1] = s;
        else
            gt[p * 3 + threadIdx.x - 10] = s;
    }
}

// one block of 256 per frame: wave w sums the outputs o = w (mod 4), four independent accumulators each (the loads of a
// single dependent chain over 412 outputs were the kernel's whole time: 61 us), then the waves' partial sums meet in LDS
__global__ __launch_bounds__(256) void k_jacobian_vjp(const float* __restrict__ jac, const float* __restrict__ g, int n_frames, int n_out, int n_in,
                                                      float* __restrict__ out) {
    const int f = blockIdx.x, k = threadIdx.x & 63, w = threadIdx.x >> 6;
    __shared__ float part[4][64];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (f < n_frames && k < n_in) {
        const float* J = jac + (size_t)f * n_out * n_in;
        const float* gg = g + (size_t)f * n_out;
        int o = w;
        for (; o + 12 < n_out; o += 16) {
            a0 = fmaf(J[(size_t)o * n_in + k], gg[o], a0);
            a1 = fmaf(J[(size_t)(o + 4) * n_in + k], gg[o + 4], a1);
            a2 = fmaf(J[(size_t)(o + 8) * n_in + k], gg[o + 8], a2);
            a3 = fmaf(J[(size_t)(o + 12) * n_in + k], gg[o + 12], a3);
        }
        for (; o < n_out; o += 4) a0 = fmaf(J[(size_t)o * n_in + k], gg[o], a0);
    }
    part[w][k] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (f < n_frames && w == 0 && k < n_in) out[(size_t)f * n_in + k] = (part[0][k] + part[1][k]) + (part[2][k] + part[3][k]);
}

// The backward pass of the whole pose side of a fitting step as ONE launch (it was a cat, two k_jacobian_vjp, a cat and a copy):
//   out[f, 0:36]  = [g_bt_inv | g_joint_3d] . jac_h[f]            (hn_pose_chain's Jacobian [399, 36])
//   out[f, 36:45] = [g_obj_r (+ g_or2) | g_obj_t (+ g_ot2)] . jac_o[f, 399:411, 0:9]  (hn_rigid_pose's Jacobian [412, 18], object half)
// Missing upstream gradients (NULL) count as zero; `which` bit 0: the hand columns, bit 1: the object columns (a pipelined step forms
// them on two streams).  One block of 256 per frame, the split of k_jacobian_vjp.
__global__ __launch_bounds__(256) void k_pose_side_vjp(const float* __restrict__ jac_h, const float* __restrict__ jac_o, const float* __restrict__ g_bt,
                                                       const float* __restrict__ g_j3, const float* __restrict__ g_or, const float* __restrict__ g_ot,
                                                       const float* __restrict__ g_or2, const float* __restrict__ g_ot2, int n_frames, int which,
                                                       float* __restrict__ out) {
    const int f = blockIdx.x, k = threadIdx.x & 63, w = threadIdx.x >> 6;
    __shared__ float part[4][64];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (f < n_frames && k < 36 && (which & 1)) {
        const float* J = jac_h + (size_t)f * 399 * 36;
        auto gg = [&](int o) { return o < 336 ? (g_bt != nullptr ? g_bt[(size_t)f * 336 + o] : 0.f) : (g_j3 != nullptr ? g_j3[(size_t)f * 63 + o - 336] : 0.f); };
        int o = w;
        for (; o + 12 < 399; o += 16) {
            a0 = fmaf(J[(size_t)o * 36 + k], gg(o), a0);
            a1 = fmaf(J[(size_t)(o + 4) * 36 + k], gg(o + 4), a1);
            a2 = fmaf(J[(size_t)(o + 8) * 36 + k], gg(o + 8), a2);
            a3 = fmaf(J[(size_t)(o + 12) * 36 + k], gg(o + 12), a3);
        }
        for (; o < 399; o += 4) a0 = fmaf(J[(size_t)o * 36 + k], gg(o), a0);
    } else if (f < n_frames && k >= 36 && k < 45 && w == 0 && (which & 2)) {
        const float* J = jac_o + (size_t)f * 412 * 18 + (size_t)399 * 18 + (k - 36);
        for (int o = 0; o < 12; ++o) {
            float g = o < 9 ? (g_or != nullptr ? g_or[(size_t)f * 9 + o] : 0.f) : (g_ot != nullptr ? g_ot[(size_t)f * 3 + o - 9] : 0.f);
            g += o < 9 ? (g_or2 != nullptr ? g_or2[(size_t)f * 9 + o] : 0.f) : (g_ot2 != nullptr ? g_ot2[(size_t)f * 3 + o - 9] : 0.f);
            a0 = fmaf(J[(size_t)o * 18], g, a0);
        }
    }
    part[w][k] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (f < n_frames && w == 0 && ((k < 36 && (which & 1)) || (k >= 36 && k < 45 && (which & 2))))
        out[(size_t)f * 45 + k] = (part[0][k] + part[1][k]) + (part[2][k] + part[3][k]);
}

// ---- the window's rows of the six pose leaves of a fitting_video sequence <-> the chain's input blocks (one launch each way) ------
// leaves (fitting_video.py:159-176): obj_rot [n,6], obj_trans [n,3], palm_rot [n,6], palm_trans [n,3], joint_refine_angle [n,20],
// palm_refine_angle [n,7].  gather: prm_h [F,36] = [joint 20 | palm_angle 7 | palm_rot 6 | palm_trans 3] (hn_pose_chain's inputs),
// prm_o [F,18] = [obj_rot 6 | obj_trans 3 | 0 x 9] (hn_rigid_pose's) of rows[f].  scatter: g [F,45] = [hand 36 | obj 9] -> the rows of
// six CONTIGUOUS gradient blocks laid out one behind the other in `out` (n x 45 floats, zeroed by the caller): [n,6] [n,3] [n,6] [n,3]
// [n,20] [n,7].  (As torch operators: a 7-tensor cat, an index_select and two slices forward; a zero fill, an index_copy and six
// strided copies backward.)
struct LeafPtrs {
    const float* p[6];
};
__global__ void k_leaf_rows_gather(LeafPtrs L, const long long* __restrict__ rows, int F, int n, float* __restrict__ prm_h, float* __restrict__ prm_o) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F * 54) return;
    const int f = i / 54, c = i % 54;
    const long long r = rows[f];
    float v = 0.f;
    if (r < 0 || r >= n)                       // a row the leaves do not have: never dereferenced, and loud (NaN) in the step's losses
        v = c < 45 ? __builtin_nanf("") : 0.f;
    else if (c < 20)
        v = L.p[4][r * 20 + c];
    else if (c < 27)
        v = L.p[5][r * 7 + c - 20];
    else if (c < 33)
        v = L.p[2][r * 6 + c - 27];
    else if (c < 36)
        v = L.p[3][r * 3 + c - 33];
    else if (c < 42)
        v = L.p[0][r * 6 + c - 36];
    else if (c < 45)
        v = L.p[1][r * 3 + c - 42];
    if (c < 36)
        prm_h[f * 36 + c] = v;
    else
        prm_o[f * 18 + c - 36] = v;
}
__global__ void k_leaf_rows_scatter(const float* __restrict__ g, const long long* __restrict__ rows, int F, int n, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F * 45) return;
    const int f = i / 45, c = i % 45;
    const long long r = rows[f];
    if (r < 0 || r >= n) return;               // (no such row: nothing is written)
    const float v = g[i];
    // block offsets in `out`: obj_rot 0, obj_trans 6 n, palm_rot 9 n, palm_trans 15 n, joint 18 n, palm_angle 38 n
    size_t o;
    if (c < 20)
        o = (size_t)18 * n + r * 20 + c;
    else if (c < 27)
        o = (size_t)38 * n + r * 7 + c - 20;
    else if (c < 33)
        o = (size_t)9 * n + r * 6 + c - 27;
    else if (c < 36)
        o = (size_t)15 * n + r * 3 + c - 33;
    else if (c < 42)
        o = r * 6 + c - 36;
    else
        o = (size_t)6 * n + r * 3 + c - 42;
    out[o] = v;
}

int rigid_pose(const float* bt_inv0, const float* joints0, const float* Ro_pred, const float* To_pred, const float* params, int n_frames, int with_palm,
               float* out, float* jac, hipStream_t s) {
    if (n_frames <= 0) return HN_OK;
    HN_REQUIRE(Ro_pred != nullptr && To_pred != nullptr && params != nullptr && out != nullptr, "rigid pose: NULL argument");
    HN_REQUIRE(!with_palm || (bt_inv0 != nullptr && joints0 != nullptr), "rigid pose: the palm half needs bt_inv0 and joints0");
    hipLaunchKernelGGL(k_rigid_pose, dim3(n_frames), dim3(32), 0, s, bt_inv0, joints0, Ro_pred, To_pred, params, n_frames, with_palm, out, jac);
    HN_LAUNCH_CHECK();
    return HN_OK;
}
int verts_loss(const float* Ra, const float* ta, const float* Rb, const float* tb, const float* verts, int n_verts, int n_pairs, float* loss, float* gR,
               float* gt, hipStream_t s) {
    if (n_pairs <= 0) return HN_OK;
    HN_REQUIRE(n_verts >= 1, "verts loss: no vertices");
    hipLaunchKernelGGL(k_verts_loss, dim3(n_pairs), dim3(256), 0, s, Ra, ta, Rb, tb, verts, n_verts, loss, gR, gt);
    HN_LAUNCH_CHECK();
    return HN_OK;
}
int pose_side_vjp(const float* jac_h, const float* jac_o, const float* g_bt, const float* g_j3, const float* g_or, const float* g_ot, const float* g_or2,
                  const float* g_ot2, int n_frames, int which, float* out, hipStream_t s) {
    if (n_frames <= 0) return HN_OK;
    HN_REQUIRE(out != nullptr && (which & 3) != 0 && (!(which & 1) || jac_h != nullptr) && (!(which & 2) || jac_o != nullptr),
               "pose_side_vjp: NULL argument");
    hipLaunchKernelGGL(k_pose_side_vjp, dim3(n_frames), dim3(256), 0, s, jac_h, jac_o, g_bt, g_j3, g_or, g_ot, g_or2, g_ot2, n_frames, which, out);
    HN_LAUNCH_CHECK();
    return HN_OK;
}
int leaf_rows_gather(const float* const* leaves6, const long long* rows, int F, int n, float* prm_h, float* prm_o, hipStream_t s) {
    if (F <= 0) return HN_OK;
    HN_REQUIRE(leaves6 != nullptr && rows != nullptr && prm_h != nullptr && prm_o != nullptr && n >= 1, "leaf_rows_gather: NULL argument");
    LeafPtrs L;
    for (int i = 0; i < 6; ++i) L.p[i] = leaves6[i];
    hipLaunchKernelGGL(k_leaf_rows_gather, dim3((F * 54 + 255) / 256), dim3(256), 0, s, L, rows, F, n, prm_h, prm_o);
    HN_LAUNCH_CHECK();
    return HN_OK;
}
int leaf_rows_scatter(const float* g, const long long* rows, int F, int n, float* out, hipStream_t s) {
    if (F <= 0) return HN_OK;
    HN_REQUIRE(g != nullptr && rows != nullptr && out != nullptr && n >= 1, "leaf_rows_scatter: NULL argument");
    hipLaunchKernelGGL(k_leaf_rows_scatter, dim3((F * 45 + 255) / 256), dim3(256), 0, s, g, rows, F, n, out);
    HN_LAUNCH_CHECK();
    return HN_OK;
}
int jacobian_vjp(const float* jac, const float* g, int n_frames, int n_out, int n_in, float* out, hipStream_t s) {
    if (n_frames <= 0) return HN_OK;
    HN_REQUIRE(n_in >= 1 && n_in <= 64 && n_out >= 1, "jacobian vjp: n_in %d n_out %d out of range", n_in, n_out);
    hipLaunchKernelGGL(k_jacobian_vjp, dim3(n_frames), dim3(256), 0, s, jac, g, n_frames, n_out, n_in, out);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

// ---- Adam over the handful of small pose-parameter blocks of a fitting loop, one launch ------------------------------------
// torch.optim.Adam's update (defaults: no weight decay, no amsgrad), written as torch writes it:
//   m += (g - m) (1 - b1);  v = b2 v + (1 - b2) g g;  p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps).
// The reference steps six blocks with six learning rates (fitting_single.py:191-199); as a torch optimiser that is one
// fused launch PER parameter group plus the step counters: 12 - 13 dependent launches of a few microseconds of work.
struct AdamArgs {
    float* p[16];
    const float* g[16];
    float* m[16];
    float* v[16];
    int n[16];
    float lr[16];
    int n_tensors;
    float bc1[16], bc2_sqrt[16];   // per block: torch keeps one step count per parameter (a block without a gradient is skipped)
    float b1, b2, eps;
};
// (one grid-stride pass over the blocks laid end to end -- element e belongs to the block whose range holds it -- so that a launch over a
// network's tensors, 16 of up to 355 k floats, does not walk them one behind the other: 19 -> ~7 us per launch of a training iteration)
__global__ __launch_bounds__(256) void k_pose_adam(const AdamArgs a) {
    int total = 0;
    for (int t = 0; t < a.n_tensors; ++t) total += a.n[t];
    for (int e = threadIdx.x + blockIdx.x * blockDim.x; e < total; e += blockDim.x * gridDim.x) {
        int t = 0, i = e;
        while (i >= a.n[t]) {
            i -= a.n[t];
            ++t;
        }
        const float step_size = a.lr[t] / a.bc1[t];
        const float bc2s = a.bc2_sqrt[t];
        const float g = a.g[t][i];
        float m = a.m[t][i], v = a.v[t][i];
        m = m + (g - m) * (1.f - a.b1);
        v = v * a.b2 + (1.f - a.b2) * g * g;
        a.m[t][i] = m;
        a.v[t][i] = v;
        const float denom = sqrtf(v) / bc2s + a.eps;
        a.p[t][i] = a.p[t][i] - step_size * (m / denom);
    }
}
int adam_step(int n_tensors, float* const* p, const float* const* g, float* const* m, float* const* v, const int* sizes, const float* lr,
              float beta1, float beta2, float eps, const int* steps, hipStream_t s) {
    if (n_tensors <= 0) return HN_OK;
    HN_REQUIRE(n_tensors <= 16 && p && g && m && v && sizes && lr && steps, "adam_step: at most 16 blocks");
    AdamArgs a{};
    int total = 0;
    for (int t = 0; t < n_tensors; ++t) {
        HN_REQUIRE(p[t] && g[t] && m[t] && v[t] && sizes[t] >= 0 && steps[t] >= 1, "adam_step: null block %d or step < 1", t);
        a.bc1[t] = (float)(1.0 - pow((double)beta1, (double)steps[t]));
        a.bc2_sqrt[t] = (float)sqrt(1.0 - pow((double)beta2, (double)steps[t]));
        a.p[t] = p[t];
        a.g[t] = g[t];
        a.m[t] = m[t];
        a.v[t] = v[t];
        a.n[t] = sizes[t];
        a.lr[t] = lr[t];
        total += sizes[t];
    }
    a.n_tensors = n_tensors;
    a.b1 = beta1;
    a.b2 = beta2;
    a.eps = eps;
    // (the pose leaves: a few hundred floats; the networks of a training iteration: up to ~1 M per launch)
    const int blocks = total > 262144 ? 256 : total > 65536 ? 64 : (total + 1023) / 1024 > 0 ? (total + 1023) / 1024 : 1;
    hipLaunchKernelGGL(k_pose_adam, dim3(blocks), dim3(256), 0, s, a);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

}  // namespace hn
