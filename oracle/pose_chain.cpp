// CPU oracle of the hand pose chain (TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may use it).  The chain is the scalar-generic restatement of fitting_single.py:206-226 and the
// halo_util functions it calls, ho-nerf_amd/csrc/hn_pose_chain.h (every function cites its reference lines), compiled
// here for the host in double precision; the device kernel instantiates the same header in float.  Because the two share
// that header, agreement between them says nothing about the restatement itself: both are pinned against
// tests/golden/pose_chain.npz -- values, refined joints and the full Jacobian obtained by executing the reference's own
// statements under autograd (tests/golden/make_golden.py, pose_goldens).
//   g++ -O2 -shared -fPIC -I ../ho-nerf_amd/csrc pose_chain.cpp -o _build/libpose_chain_oracle.so   (oracle/Makefile)
#include "hn_pose_chain.h"

using hn::pose::Dual;
using hn::pose::N_IN;
using hn::pose::N_OUT;

// the per-finger form of the chain (what the device kernel runs) against the whole-hand form: max |difference| over all
// outputs and directions of the given frames
extern "C" double oracle_pose_chain_forms_differ(const double* ori_pose, const double* bone_len, const double* params, int n_frames) {
    double worst = 0.0;
    for (int f = 0; f < n_frames; ++f) {
        double pose[21][3], bl[20];
        for (int i = 0; i < 63; ++i) pose[i / 3][i % 3] = ori_pose[f * 63 + i];
        for (int i = 0; i < 20; ++i) bl[i] = bone_len[f * 20 + i];
        for (int k = 0; k <= N_IN; ++k) {
            Dual<double> x[N_IN], y[N_OUT], z[N_OUT];
            for (int i = 0; i < N_IN; ++i) x[i] = Dual<double>(params[f * N_IN + i], i == k - 1 ? 1.0 : 0.0);
            hn::pose::pose_chain<double>(pose, bl, true, x, y);
            hn::pose::pose_chain_by_finger<double>(pose, bl, true, x, z);
            for (int i = 0; i < N_OUT; ++i) {
                const double dv = y[i].v - z[i].v, dd = y[i].d - z[i].d;
                if (dv < 0 ? -dv > worst : dv > worst) worst = dv < 0 ? -dv : dv;
                if (dd < 0 ? -dd > worst : dd > worst) worst = dd < 0 ? -dd : dd;
            }
        }
    }
    return worst;
}

extern "C" int oracle_pose_chain(const double* ori_pose /* [F,21,3] */, const double* bone_len /* [F,20] */, int is_right,
                                 const double* params /* [F,36] */, int n_frames, double* bt_inv /* [F,21,16] */,
                                 double* joint_3d /* [F,21,3] */, double* jac /* [F,399,36] or NULL */) {
    for (int f = 0; f < n_frames; ++f) {
        double pose[21][3], bl[20];
        for (int i = 0; i < 63; ++i) pose[i / 3][i % 3] = ori_pose[f * 63 + i];
        for (int i = 0; i < 20; ++i) bl[i] = bone_len[f * 20 + i];
        for (int k = 0; k <= (jac ? N_IN : 0); ++k) {
            Dual<double> x[N_IN], y[N_OUT];
            for (int i = 0; i < N_IN; ++i) x[i] = Dual<double>(params[f * N_IN + i], i == k - 1 ? 1.0 : 0.0);
            hn::pose::pose_chain<double>(pose, bl, is_right != 0, x, y);
            if (k == 0) {
                for (int i = 0; i < 336; ++i) bt_inv[f * 336 + i] = y[i].v;
                for (int i = 0; i < 63; ++i) joint_3d[f * 63 + i] = y[336 + i].v;
            } else {
                for (int i = 0; i < N_OUT; ++i) jac[((size_t)f * N_OUT + i) * N_IN + (k - 1)] = y[i].d;
            }
        }
    }
    return 0;
}
