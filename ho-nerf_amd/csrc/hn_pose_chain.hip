// hn_pose_chain / hn_pose_chain_bwd: the hand pose chain of the fitting loops (fitting_single.py:206-226 and the
// halo_util functions it calls, ~4 000 torch operators per step in the reference) as two launches.  The chain itself is
// hn_pose_chain.h; here one thread per (frame, input direction, finger) evaluates it on dual numbers: direction 0 writes
// the values, direction 1 + k the column d(outputs)/d(input k) of the Jacobian [N_OUT x N_IN], which the backward launch
// contracts with the upstream gradient.  The work is a few kFLOP per thread; the point is the launch count.
#include "hn_common.h"
#include "hn_pose_chain.h"

namespace hn {

using pose::Dual;
using pose::N_IN;
using pose::N_OUT;

// One block per frame, one thread per (input direction k, finger f): 37 x 5 = 185 threads.  The chain's four phases
// (hn_pose_chain.h, "one FINGER at a time") with three exchanges through LDS between them.
// VALUES: the outputs alone (jac == NULL: 5 threads, one per finger).  No tangent is published, fetched or stored, so every
// tangent operation of the dual arithmetic is dead code there and the chain is a third of the instructions: the values are what
// a fitting step's render waits for, the Jacobian is not needed before the step's backward pass reaches the pose side, and
// the pipelined step (fitting.PipelinedSingleFit) asks for the two in separate launches (bt_inv == NULL: Jacobian only).
constexpr int PC_DIRS = N_IN + 1;
template <bool VALUES>
__global__ __launch_bounds__(VALUES ? 64 : 192) void k_pose_chain(const float* __restrict__ ori_pose, const float* __restrict__ bone_len,
                                                                   const unsigned char* __restrict__ is_right, const float* __restrict__ in,
                                                                   int n_frames, float* __restrict__ bt_inv, float* __restrict__ joint_3d,
                                                                   float* __restrict__ jac) {
    using S = std::conditional_t<VALUES, pose::VD, double>;   // VD: the chain without tangents (hn_pose_chain.h)
    using T = Dual<S>;
    using pose::V3;
    constexpr int DIRS = VALUES ? 1 : PC_DIRS;
    __shared__ double xb[DIRS][5][VALUES ? 3 : 6];
    const int fr = blockIdx.x, tid = threadIdx.x;
    const int k = tid % DIRS, f = tid / DIRS;   // k = 0: values; 1 + input index: that input's derivative
    const bool active = fr < n_frames && f < 5;
    // In double: in fp32 the chain's angle / normalisation steps leave 3e-4 of relative error on the Jacobian (measured against
    // the reference's fp64 run; the reference's own fp32 run is 3e-5 off on the values).  Inputs and outputs stay fp32.
    S pose[21][3], bl[20];
    pose::ChainIn<S> ci;
    pose::FingerState<T> st;
    if (active) {
        for (int i = 0; i < 63; ++i) pose[i / 3][i % 3] = S((double)ori_pose[(size_t)fr * 63 + i]);
        for (int i = 0; i < 20; ++i) bl[i] = S((double)bone_len[(size_t)fr * 20 + i]);
        T x[N_IN];
        for (int i = 0; i < N_IN; ++i) x[i] = T(S((double)in[(size_t)fr * N_IN + i]), S((!VALUES && i == k - 1) ? 1.0 : 0.0));
        pose::chain_inputs<S>(pose, bl, is_right == nullptr || is_right[fr] != 0, x, ci);
    }
    auto publish = [&](const V3<T>& v) {
        for (int c = 0; c < 3; ++c) {
            if constexpr (VALUES) {
                xb[k][f][c] = (double)v.x[c].v;
            } else {
                xb[k][f][2 * c] = v.x[c].v;
                xb[k][f][2 * c + 1] = v.x[c].d;
            }
        }
    };
    auto fetch = [&](int g) {
        V3<T> v;
        for (int c = 0; c < 3; ++c) {
            if constexpr (VALUES)
                v.x[c] = T(S(xb[k][g][c]));
            else
                v.x[c] = T(S(xb[k][g][2 * c]), S(xb[k][g][2 * c + 1]));
        }
        return v;
    };
    V3<T> RB[5];
    if (active) publish(pose::chain_phase_a<S>(f, ci, st));
    __syncthreads();
    if (active)
        for (int g = 0; g < 5; ++g) RB[g] = fetch(g);
    __syncthreads();
    if (active) publish(pose::chain_phase_b<S>(f, ci, RB, st));
    __syncthreads();
    V3<T> J1, J2;
    if (active) {
        J1 = fetch(1);
        J2 = fetch(2);
    }
    __syncthreads();
    if (active) publish(pose::chain_phase_c<S>(f, ci, J1, J2, st));
    __syncthreads();
    if (active) {
        for (int g = 0; g < 5; ++g) RB[g] = fetch(g);
        float* bt = bt_inv != nullptr ? bt_inv + (size_t)fr * 336 : nullptr;
        float* j3 = joint_3d != nullptr ? joint_3d + (size_t)fr * 63 : nullptr;
        float* J = jac != nullptr ? jac + (size_t)fr * N_OUT * N_IN : nullptr;
        pose::chain_phase_d<S>(f, ci, RB, st, [&](int idx, const T& v) {
            if (k == 0) {
                if (bt != nullptr) {
                    if (idx < 336)
                        bt[idx] = (float)v.v;
                    else
                        j3[idx - 336] = (float)v.v;
                }
            } else if constexpr (!VALUES) {
                J[(size_t)idx * N_IN + (k - 1)] = (float)v.d;
            }
        });
    }
}

// g_in[f][k] = sum_o jac[f][o][k] g_out[f][o];  g_out = [g_bt_inv 336 | g_joint_3d 63] (either may be NULL = zero)
__global__ __launch_bounds__(64) void k_pose_chain_bwd(const float* __restrict__ jac, const float* __restrict__ g_bt, const float* __restrict__ g_j3,
                                                       int n_frames, float* __restrict__ g_in) {
    const int f = blockIdx.x, k = threadIdx.x;
    if (f >= n_frames || k >= N_IN) return;
    const float* J = jac + (size_t)f * N_OUT * N_IN;
    float acc = 0.f;
    if (g_bt != nullptr)
        for (int o = 0; o < 336; ++o) acc = fmaf(J[o * N_IN + k], g_bt[(size_t)f * 336 + o], acc);
    if (g_j3 != nullptr)
        for (int o = 0; o < 63; ++o) acc = fmaf(J[(336 + o) * N_IN + k], g_j3[(size_t)f * 63 + o], acc);
    g_in[(size_t)f * N_IN + k] = acc;
}

int pose_chain(const float* ori_pose, const float* bone_len, const unsigned char* is_right, const float* in, int n_frames, float* bt_inv,
               float* joint_3d, float* jac, hipStream_t s) {
    if (n_frames <= 0) return HN_OK;
    HN_REQUIRE(ori_pose != nullptr && bone_len != nullptr && in != nullptr, "pose chain: NULL argument");
    HN_REQUIRE((bt_inv != nullptr) == (joint_3d != nullptr), "pose chain: bt_inv and joint_3d are written together (both or neither)");
    HN_REQUIRE(bt_inv != nullptr || jac != nullptr, "pose chain: nothing to compute (no value outputs and no Jacobian)");
    if (jac == nullptr)
        hipLaunchKernelGGL(k_pose_chain<true>, dim3(n_frames), dim3(64), 0, s, ori_pose, bone_len, is_right, in, n_frames, bt_inv, joint_3d, jac);
    else
        hipLaunchKernelGGL(k_pose_chain<false>, dim3(n_frames), dim3(192), 0, s, ori_pose, bone_len, is_right, in, n_frames, bt_inv, joint_3d, jac);
    HN_LAUNCH_CHECK();
    return HN_OK;
}
int pose_chain_bwd(const float* jac, const float* g_bt_inv, const float* g_joint_3d, int n_frames, float* g_in, hipStream_t s) {
    if (n_frames <= 0) return HN_OK;
    HN_REQUIRE(jac != nullptr && g_in != nullptr, "pose chain backward: NULL argument");
    hipLaunchKernelGGL(k_pose_chain_bwd, dim3(n_frames), dim3(64), 0, s, jac, g_bt_inv, g_joint_3d, n_frames, g_in);
    HN_LAUNCH_CHECK();
    return HN_OK;
}

}  // namespace hn
