// Sanitizer harness (CPU): the hand pose chain header of the product (ho-nerf_amd/csrc/hn_pose_chain.h, what k_pose_chain
// instantiates in float) through the double-precision entry points of oracle/pose_chain.cpp, compiled with
// -fsanitize=address,undefined: values + Jacobian of a few deterministic hands, and the per-finger form (what the device
// kernel runs) against the whole-hand form.  TEST INFRASTRUCTURE (tests/san/Makefile, tests/test_abi.py).
#include <math.h>
#include <stdio.h>

#include <vector>

extern "C" double oracle_pose_chain_forms_differ(const double*, const double*, const double*, int);
extern "C" int oracle_pose_chain(const double*, const double*, int, const double*, int, double*, double*, double*);

int main() {
    const int F = 3;
    std::vector<double> pose(F * 63), bl(F * 20), prm(F * 36), bt(F * 336), j3(F * 63), jac((size_t)F * 399 * 36);
    // a flat open hand in MANO joint order (wrist, then four joints per finger along +x, fingers fanned in y), perturbed
    unsigned s = 12345u;
    auto rnd = [&]() {
        s = s * 1664525u + 1013904223u;
        return (double)(s >> 8) / (double)(1u << 24) - 0.5;
    };
    for (int f = 0; f < F; ++f) {
        double* p = &pose[f * 63];
        p[0] = p[1] = p[2] = 0.0;
        for (int fi = 0; fi < 5; ++fi)
            for (int k = 0; k < 4; ++k) {
                const int j = 1 + 4 * fi + k;
                p[3 * j] = 0.03 + 0.025 * (k + 1) + 0.004 * rnd();
                p[3 * j + 1] = 0.02 * (fi - 2) * (1.0 + 0.3 * k) + 0.004 * rnd();
                p[3 * j + 2] = 0.01 * rnd() - 0.004 * k * (fi == 0 ? 2.0 : 1.0);
            }
        for (int i = 0; i < 20; ++i) bl[f * 20 + i] = 0.02 + 0.03 * (rnd() + 0.5);
        for (int i = 0; i < 36; ++i) prm[f * 36 + i] = 0.05 * rnd();
        const double eye62[6] = {1, 0, 0, 1, 0, 0};
        for (int i = 0; i < 6; ++i) prm[f * 36 + 27 + i] += eye62[i];
    }
    if (oracle_pose_chain(pose.data(), bl.data(), 1, prm.data(), F, bt.data(), j3.data(), jac.data()) != 0) return 2;
    int bad = 0;
    for (double v : bt) bad += !std::isfinite(v);
    for (double v : j3) bad += !std::isfinite(v);
    for (double v : jac) bad += !std::isfinite(v);
    const double d = oracle_pose_chain_forms_differ(pose.data(), bl.data(), prm.data(), F);
    printf("pose chain: %d non-finite outputs; per-finger form vs whole-hand form: max |difference| %.3e\n", bad, d);
    if (bad || !(d < 1e-9)) {
        printf("FAILED\n");
        return 1;
    }
    printf("pose chain: ok\n");
    return 0;
}
