// The hand pose chain of the fitting loops as one scalar-generic function: refine angles + palm rotation / translation
// -> refined 3-D joints -> the 21 inverse bone transformations the hand field takes (`bone_transformation_inv`).
//
// Restates, statement by statement, what the reference runs per optimisation step as ~4 000 small torch operators:
//   fitting_single.py:206-226 (the same lines in fitting_video.py)           the chain itself
//   halo_util/converter_fit_batch.py:103-162   transform_to_canonical / compute_canonical_transform
//   halo_util/converter_fit_batch.py:1183-1229 PoseConverter.get_refine_3d_joint
//   halo_util/converter_fit_batch.py:1109-1179 PoseConverter.forward
//   ... :537-562 kp3D_to_bones, :564-594 compute_bone_to_kp_mat, :596-722 compute_local_coordinate_system (detached),
//   :731-766 compute_rot_angles, :769-808 preprocess_joints, :811-875 compute_rotation_matrix,
//   :939-962 compute_adjusted_transpose, :964-1031 normalize_root_planes, :1033-1107 normalize_root_bone_angles
//   halo_util/utils.py:17-41 convert_joints (index tables), utils/utils.py:11-30 rot6d_to_matrix
//
// Derivatives come from forward-mode dual numbers: the function is written once over Dual<S>, a caller seeds one input
// direction and reads d(outputs)/d(that input).  The device kernel (hn_pose_chain.hip) runs one thread per (frame, input);
// the CPU oracle (oracle/pose_chain.cpp) compiles this same header for the host -- both with S = double, both pinned
// against tests/golden/pose_chain.npz, which is produced by executing the reference's own statements (make_golden.py).
// `detach` (the reference's .detach() / .clone().detach()) drops the tangent.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define HN_PC_FN __host__ __device__ inline
#else
#define HN_PC_FN inline
#endif

namespace hn {
namespace pose {

constexpr int N_JRA = 20, N_PRA = 7, N_ROT6 = 6, N_TRANS = 3;
constexpr int N_IN = N_JRA + N_PRA + N_ROT6 + N_TRANS;   // 36 differentiable inputs, in this order
constexpr int N_OUT = 21 * 16 + 21 * 3;                    // bone_transformation_inv [21][4][4], joint_3d [21][3]

template <typename S>
struct Dual {
    S v, d;
    HN_PC_FN Dual() : v(0), d(0) {}
    HN_PC_FN Dual(S a) : v(a), d(0) {}
    HN_PC_FN Dual(S a, S b) : v(a), d(b) {}
};
// ---- the chain WITHOUT tangents: S = VD ("value double": a double in a struct, the same arithmetic) selects the specialisation
// Dual<VD> whose tangent is a Zero -- an empty type whose operations do nothing -- so that the generic dual arithmetic below
// compiles to the values alone.  (With S = double and all tangents zero the compiler cannot drop them: they pass through the
// fingers' state arrays in scratch memory.)  hn_pose_chain with jac == NULL runs this instantiation: 0.09 -> 0.0x ms on the
// critical path of every fitting step.
struct VD {
    double x;
    HN_PC_FN VD() : x(0.0) {}
    HN_PC_FN VD(double a) : x(a) {}
    HN_PC_FN explicit operator double() const { return x; }
    HN_PC_FN explicit operator float() const { return (float)x; }
};
HN_PC_FN VD operator+(VD a, VD b) { return VD(a.x + b.x); }
HN_PC_FN VD operator-(VD a, VD b) { return VD(a.x - b.x); }
HN_PC_FN VD operator*(VD a, VD b) { return VD(a.x * b.x); }
HN_PC_FN VD operator/(VD a, VD b) { return VD(a.x / b.x); }
HN_PC_FN VD operator-(VD a) { return VD(-a.x); }
HN_PC_FN VD& operator+=(VD& a, VD b) { a.x += b.x; return a; }
HN_PC_FN VD& operator-=(VD& a, VD b) { a.x -= b.x; return a; }
HN_PC_FN VD& operator*=(VD& a, VD b) { a.x *= b.x; return a; }
HN_PC_FN bool operator<(VD a, VD b) { return a.x < b.x; }
HN_PC_FN bool operator>(VD a, VD b) { return a.x > b.x; }
HN_PC_FN bool operator<=(VD a, VD b) { return a.x <= b.x; }
HN_PC_FN bool operator>=(VD a, VD b) { return a.x >= b.x; }
HN_PC_FN bool operator==(VD a, VD b) { return a.x == b.x; }
HN_PC_FN bool operator!=(VD a, VD b) { return a.x != b.x; }
struct Zero {
    HN_PC_FN Zero() {}
    HN_PC_FN Zero(VD) {}      // (`cond ? tangent expression : S(0)` of the generic code)
};
HN_PC_FN Zero operator+(Zero, Zero) { return {}; }
HN_PC_FN Zero operator-(Zero, Zero) { return {}; }
HN_PC_FN Zero operator-(Zero) { return {}; }
HN_PC_FN Zero operator*(Zero, VD) { return {}; }
HN_PC_FN Zero operator*(VD, Zero) { return {}; }
HN_PC_FN Zero operator/(Zero, VD) { return {}; }
template <>
struct Dual<VD> {
    VD v;
    Zero d;
    HN_PC_FN Dual() : v(0.0) {}
    HN_PC_FN Dual(VD a) : v(a) {}
    HN_PC_FN Dual(VD a, Zero) : v(a) {}
    HN_PC_FN Dual(VD a, VD) : v(a) {}
};
template <typename S> HN_PC_FN Dual<S> operator+(Dual<S> a, Dual<S> b) { return {a.v + b.v, a.d + b.d}; }
template <typename S> HN_PC_FN Dual<S> operator-(Dual<S> a, Dual<S> b) { return {a.v - b.v, a.d - b.d}; }
template <typename S> HN_PC_FN Dual<S> operator-(Dual<S> a) { return {-a.v, -a.d}; }
template <typename S> HN_PC_FN Dual<S> operator*(Dual<S> a, Dual<S> b) { return {a.v * b.v, a.d * b.v + a.v * b.d}; }
template <typename S> HN_PC_FN Dual<S> operator/(Dual<S> a, Dual<S> b) { return {a.v / b.v, (a.d - a.v / b.v * b.d) / b.v}; }
template <typename S> HN_PC_FN Dual<S> operator*(S a, Dual<S> b) { return {a * b.v, a * b.d}; }
template <typename S> HN_PC_FN Dual<S> operator*(Dual<S> a, S b) { return {a.v * b, a.d * b}; }
template <typename S> HN_PC_FN Dual<S> operator+(Dual<S> a, S b) { return {a.v + b, a.d}; }
template <typename S> HN_PC_FN Dual<S> operator-(Dual<S> a, S b) { return {a.v - b, a.d}; }
template <typename S> HN_PC_FN Dual<S> operator-(S a, Dual<S> b) { return {a - b.v, -b.d}; }
template <typename S> HN_PC_FN Dual<S> detach(Dual<S> a) { return {a.v, S(0)}; }
HN_PC_FN float sqrt_s(float x) { return sqrtf(x); }
HN_PC_FN double sqrt_s(double x) { return sqrt(x); }
HN_PC_FN float sin_s(float x) { return sinf(x); }
HN_PC_FN double sin_s(double x) { return sin(x); }   // (the chain itself goes through sincos_s)
HN_PC_FN float cos_s(float x) { return cosf(x); }
HN_PC_FN double cos_s(double x) { return cos(x); }
HN_PC_FN float acos_s(float x) { return acosf(x); }
#if defined(__HIP_DEVICE_COMPILE__)
// acos by fdlibm's rational approximation of asin (no hi / lo tail correction: ~1e-16 relative); arguments are clipped to
// [-1 + 1e-6, 1 - 1e-6] by the chain before they get here
HN_PC_FN double acos_s(double x) {
    auto R = [](double z) {
        const double p = z * (1.66666666666666657415e-01 + z * (-3.25565818622400915405e-01 + z * (2.01212532134862925881e-01 +
                         z * (-4.00555345006794114027e-02 + z * (7.91534994289814532176e-04 + z * 3.47933107596021167570e-05)))));
        const double q = 1.0 + z * (-2.40339491173441421878e+00 + z * (2.02094576023350569471e+00 + z * (-6.88283971605453293030e-01 +
                         z * 7.70381505559019352791e-02)));
        return p / q;
    };
    if (fabs(x) < 0.5) return 1.57079632679489655800e+00 - (x + x * R(x * x));
    const double z = (1.0 - fabs(x)) * 0.5, sq = sqrt(z);
    const double a = 2.0 * (sq + sq * R(z));
    return x > 0.0 ? a : 3.14159265358979311600e+00 - a;
}
#else
HN_PC_FN double acos_s(double x) { return acos(x); }
#endif
HN_PC_FN float atan2_s(float y, float x) { return atan2f(y, x); }
HN_PC_FN double acos_s(double x);
HN_PC_FN void sincos_s(double x, double& s, double& c);
HN_PC_FN VD sqrt_s(VD a) { return VD(sqrt_s(a.x)); }
HN_PC_FN VD sin_s(VD a) { return VD(sin_s(a.x)); }
HN_PC_FN VD cos_s(VD a) { return VD(cos_s(a.x)); }
HN_PC_FN VD acos_s(VD a) { return VD(acos_s(a.x)); }
HN_PC_FN VD atan2_s(VD y, VD x);
HN_PC_FN void sincos_s(VD a, VD& s, VD& c) { sincos_s(a.x, s.x, c.x); }
HN_PC_FN double atan2_s(double y, double x) { return atan2(y, x); }
HN_PC_FN VD atan2_s(VD y, VD x) { return VD(atan2_s(y.x, x.x)); }
HN_PC_FN void sincos_s(float x, float& s, float& c) { sincosf(x, &s, &c); }
#if defined(__HIP_DEVICE_COMPILE__)
// Device build: the chain's angles are a few radians at most, and the library's double-precision sincos / acos (huge-argument
// reduction, special cases: ~200 instructions each, ~450 calls per evaluation) were most of the kernel.  Two-term Cody-Waite
// reduction by pi/2 and the fdlibm kernel polynomials: < 2 ulp for |x| < 1e4.
HN_PC_FN void sincos_s(double x, double& s, double& c) {
    const double k = rint(x * 0.63661977236758134308);
    double r = fma(-k, 1.57079632679489655800e+00, x);
    r = fma(-k, 6.12323399573676603587e-17, r);
    const double z = r * r;
    const double ps = -1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 +
                      z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
    const double pc = 4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                      z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
    const double sr = fma(r * z, ps, r), cr = fma(z * z, pc, fma(-0.5, z, 1.0));
    const int q = (int)k & 3;
    s = (q & 1) ? cr : sr;
    c = (q & 1) ? sr : cr;
    if (q == 1 || q == 2) c = -c;
    if (q >= 2) s = -s;
}
#else
HN_PC_FN void sincos_s(double x, double& s, double& c) { sincos(x, &s, &c); }
#endif
// sin and cos of one angle from one argument reduction (the double-precision library routines are the bulk of the chain's
// instructions: every angle is evaluated once and its pair handed to whatever rotates by it)
template <typename S> struct SinCos { Dual<S> s, c; };
template <typename S> HN_PC_FN SinCos<S> sincos_d(Dual<S> a) {
    S sv, cv;
    sincos_s(a.v, sv, cv);
    return {{sv, cv * a.d}, {cv, -sv * a.d}};
}
template <typename S> HN_PC_FN Dual<S> sin_d(Dual<S> a) { return {sin_s(a.v), cos_s(a.v) * a.d}; }
template <typename S> HN_PC_FN Dual<S> cos_d(Dual<S> a) { return {cos_s(a.v), -sin_s(a.v) * a.d}; }
template <typename S> HN_PC_FN Dual<S> acos_d(Dual<S> a) { return {acos_s(a.v), -a.d / sqrt_s(S(1) - a.v * a.v)}; }
template <typename S> HN_PC_FN Dual<S> atan2_d(Dual<S> y, Dual<S> x) {
    const S den = x.v * x.v + y.v * y.v;
    return {atan2_s(y.v, x.v), den > S(0) ? (x.v * y.d - y.v * x.d) / den : S(0)};
}
// torch.max(x, eps) / the two-sided clip: the tangent passes where x is the selected argument
template <typename S> HN_PC_FN Dual<S> max_c(Dual<S> a, S c) { return a.v >= c ? a : Dual<S>(c); }
template <typename S> HN_PC_FN Dual<S> clip_c(Dual<S> a, S lo, S hi) { return a.v < lo ? Dual<S>(lo) : (a.v > hi ? Dual<S>(hi) : a); }

template <typename T> struct V3 { T x[3]; };
template <typename T> struct M3 { T m[3][3]; };
template <typename T> struct M4 { T m[4][4]; };

template <typename T> HN_PC_FN V3<T> operator+(const V3<T>& a, const V3<T>& b) { return {{a.x[0] + b.x[0], a.x[1] + b.x[1], a.x[2] + b.x[2]}}; }
template <typename T> HN_PC_FN V3<T> operator-(const V3<T>& a, const V3<T>& b) { return {{a.x[0] - b.x[0], a.x[1] - b.x[1], a.x[2] - b.x[2]}}; }
template <typename T> HN_PC_FN V3<T> scale(const V3<T>& a, T s) { return {{a.x[0] * s, a.x[1] * s, a.x[2] * s}}; }
template <typename T> HN_PC_FN T dot(const V3<T>& a, const V3<T>& b) { return a.x[0] * b.x[0] + a.x[1] * b.x[1] + a.x[2] * b.x[2]; }
template <typename T> HN_PC_FN V3<T> cross(const V3<T>& a, const V3<T>& b) {
    return {{a.x[1] * b.x[2] - a.x[2] * b.x[1], a.x[2] * b.x[0] - a.x[0] * b.x[2], a.x[0] * b.x[1] - a.x[1] * b.x[0]}};
}
// torch.norm: the derivative at 0 is 0
template <typename S> HN_PC_FN Dual<S> norm(const V3<Dual<S>>& a) {
    const S n = sqrt_s(a.x[0].v * a.x[0].v + a.x[1].v * a.x[1].v + a.x[2].v * a.x[2].v);
    return {n, n > S(0) ? (a.x[0].v * a.x[0].d + a.x[1].v * a.x[1].d + a.x[2].v * a.x[2].d) / n : S(0)};
}
template <typename S> HN_PC_FN V3<Dual<S>> unit(const V3<Dual<S>>& a, S eps) { return scale(a, Dual<S>(S(1)) / max_c(norm(a), eps)); }
template <typename T> HN_PC_FN V3<T> detach3(const V3<T>& a) { return {{detach(a.x[0]), detach(a.x[1]), detach(a.x[2])}}; }
template <typename T> HN_PC_FN M3<T> eye3() {
    M3<T> r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = T(i == j ? 1 : 0);
    return r;
}
template <typename T> HN_PC_FN M3<T> mul(const M3<T>& a, const M3<T>& b) {
    M3<T> r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
    return r;
}
template <typename T> HN_PC_FN V3<T> mul(const M3<T>& a, const V3<T>& v) {
    V3<T> r;
    for (int i = 0; i < 3; ++i) r.x[i] = a.m[i][0] * v.x[0] + a.m[i][1] * v.x[1] + a.m[i][2] * v.x[2];
    return r;
}
template <typename T> HN_PC_FN M3<T> transpose(const M3<T>& a) {
    M3<T> r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[j][i];
    return r;
}
template <typename T> HN_PC_FN M3<T> detachM(const M3<T>& a) {
    M3<T> r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = detach(a.m[i][j]);
    return r;
}
// torch.inverse of a 3 x 3 matrix (adjugate / determinant)
template <typename T> HN_PC_FN M3<T> inverse(const M3<T>& a) {
    M3<T> c;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            c.m[j][i] = a.m[i1][j1] * a.m[i2][j2] - a.m[i1][j2] * a.m[i2][j1];   // cofactor (i, j) -> adjugate (j, i)
        }
    const T det = a.m[0][0] * c.m[0][0] + a.m[0][1] * c.m[1][0] + a.m[0][2] * c.m[2][0];
    M3<T> r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = c.m[i][j] / det;
    return r;
}
template <typename T> HN_PC_FN M4<T> eye4() {
    M4<T> r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r.m[i][j] = T(i == j ? 1 : 0);
    return r;
}
template <typename T> HN_PC_FN M4<T> mul(const M4<T>& a, const M4<T>& b) {
    M4<T> r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j] + a.m[i][3] * b.m[3][j];
    return r;
}
template <typename T> HN_PC_FN M4<T> to44(const M3<T>& a) {   // from_3x3_mat_to_4x4 (:931-937)
    M4<T> r = eye4<T>();
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][j];
    return r;
}

// angle2 (:62-74): 2 atan2(|n1 - n2|, |n1 + n2|)
template <typename S> HN_PC_FN Dual<S> angle2(const V3<Dual<S>>& v1, const V3<Dual<S>>& v2) {
    const V3<Dual<S>> n1 = unit(v1, S(1e-10)), n2 = unit(v2, S(1e-10));
    return S(2) * atan2_d(norm(n1 - n2), norm(n1 + n2));
}
// signed_angle (:76-92)
template <typename S> HN_PC_FN Dual<S> signed_angle(const V3<Dual<S>>& v1, const V3<Dual<S>>& v2, const V3<Dual<S>>& ref) {
    const Dual<S> a = angle2(v1, v2);
    return dot(ref, cross(v1, v2)).v < S(0) ? -a : a;
}
// rotation_matrix (:280-309): Rodrigues, the axis re-normalised (F.normalize, eps 1e-12)
template <typename S> HN_PC_FN M3<Dual<S>> rotation_matrix(Dual<S> angle, const V3<Dual<S>>& axis_in) {
    using T = Dual<S>;
    const V3<T> a = unit(axis_in, S(1e-12));
    const SinCos<S> sc = sincos_d(angle);
    const T s = sc.s, c1 = S(1) - sc.c, o = T(S(0));
    M3<T> K;
    K.m[0][0] = o; K.m[0][1] = -a.x[2]; K.m[0][2] = a.x[1];
    K.m[1][0] = a.x[2]; K.m[1][1] = o; K.m[1][2] = -a.x[0];
    K.m[2][0] = -a.x[1]; K.m[2][1] = a.x[0]; K.m[2][2] = o;
    const M3<T> K2 = mul(K, K);
    M3<T> R = eye3<T>();
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R.m[i][j] = R.m[i][j] + K.m[i][j] * s + K2.m[i][j] * c1;
    return R;
}
// rotate (:323-338) = rotate_axis_angle (:34-40): Rodrigues on a vector, the axis taken as given
template <typename S> HN_PC_FN V3<Dual<S>> rotate(const V3<Dual<S>>& v, const V3<Dual<S>>& ax, const SinCos<S>& sc) {
    return scale(v, sc.c) + scale(cross(ax, v), sc.s) + scale(ax, dot(ax, v) * (S(1) - sc.c));
}
template <typename S> HN_PC_FN V3<Dual<S>> rotate(const V3<Dual<S>>& v, const V3<Dual<S>>& ax, Dual<S> rad) {
    return rotate(v, ax, sincos_d(rad));
}
// get_alignment_mat (:94-101)
template <typename S> HN_PC_FN M3<Dual<S>> alignment_mat(const V3<Dual<S>>& v1, const V3<Dual<S>>& v2) {
    return rotation_matrix(angle2(v1, v2), unit(cross(v1, v2), S(1e-8)));
}

// index tables of convert_joints (halo_util/utils.py:18-23)
HN_PC_FN int mano_to_biomech(int i) {
    const int t[21] = {0, 1, 5, 9, 13, 17, 2, 6, 10, 14, 18, 3, 7, 11, 15, 19, 4, 8, 12, 16, 20};
    return t[i];
}
HN_PC_FN int biomech_to_mano(int i) {
    const int t[21] = {0, 1, 6, 11, 16, 2, 7, 12, 17, 3, 8, 13, 18, 4, 9, 14, 19, 5, 10, 15, 20};
    return t[i];
}

// compute_canonical_transform (:116-162) on detached joints: T [3][4] with T kp = R_2 R_1 (kp - root) (rows flipped in y for
// a left hand)
template <typename S> HN_PC_FN void canonical_transform(const V3<Dual<S>> (&kp_in)[21], bool is_right, M3<Dual<S>>& R, V3<Dual<S>>& t) {
    using T = Dual<S>;
    V3<T> k0 = detach3(kp_in[0]), k2 = detach3(kp_in[2]), k3 = detach3(kp_in[3]);
    if (!is_right) {
        k0.x[1] = -k0.x[1];
        k2.x[1] = -k2.x[1];
        k3.x[1] = -k3.x[1];
    }
    const V3<T> y_axis = {{T(S(0)), T(S(-1)), T(S(0))}}, z_axis = {{T(S(0)), T(S(0)), T(S(1))}};
    const V3<T> v_mrb = unit(k3 - k0, S(1e-8));
    const M3<T> R1 = alignment_mat(v_mrb, y_axis);
    const V3<T> v_irb = unit(k2 - k0, S(1e-8));
    const V3<T> normal = cross(v_mrb, v_irb);
    const V3<T> normal_rot = mul(R1, normal);   // normal @ R_1^T
    const M3<T> R2 = alignment_mat(normal_rot, z_axis);
    M3<T> Tt = eye3<T>();
    if (!is_right) Tt.m[1][1] = T(S(-1));
    R = mul(R2, mul(R1, Tt));
    // translation column of R_2 R_1 T_t with T_t[:, 3] = -root (the flipped root)
    t = mul(mul(R2, R1), scale(k0, T(S(-1))));
}

// kp3D_to_bones (:537-562).  kp_to_bone_mat = scale(1 / length) . translate(-parent joint) is not formed: the only place
// it is used multiplies it by scale(length) again (see converter_forward), so the parent joints are handed on instead.
template <typename S> HN_PC_FN void kp_to_bones(const V3<Dual<S>> (&kp)[21], V3<Dual<S>> (&bones)[20], Dual<S> (&bl)[20], V3<Dual<S>> (&kpar)[20]) {
    using T = Dual<S>;
    for (int i = 0; i < 20; ++i) {
        const int p = i < 5 ? 0 : i - 4;   // idx_2: the parent joint
        const V3<T> b = kp[i + 1] - kp[p];
        bl[i] = max_c(norm(b), S(1e-9));
        bones[i] = scale(b, T(S(1)) / bl[i]);
        kpar[i] = kp[p];
    }
}

// normalize_root_planes (:964-1031); pra: palm_refine_angle (entries 0..2 used)
template <typename S, int N> HN_PC_FN void normalize_root_planes(const V3<Dual<S>> (&bones)[N], const Dual<S> (&pra)[N_PRA], V3<Dual<S>> (&out)[N], M3<Dual<S>> (&mat)[N]) {
    using T = Dual<S>;
    const S canon[3] = {S(0.8), S(0.2), S(0.2)};   // self.root_plane_angles without a canonical pose (:405)
    V3<T> b0 = bones[0], b1 = bones[1], b2 = bones[2], b3 = bones[3], b4 = bones[4];
    M3<T> root[5];
    for (int i = 0; i < 5; ++i) root[i] = eye3<T>();
    const V3<T> n1 = cross(b2, b1);
    const V3<T> n0 = cross(b1, b0);
    const T a01 = signed_angle(n0, n1, b1);
    root[0] = rotation_matrix(a01 - canon[0] + pra[0], b1);
    V3<T> n2 = cross(b3, b2);
    const T a21 = signed_angle(n2, n1, b2);
    const M3<T> ring = rotation_matrix(a21 + canon[1] + pra[1], b2);
    b3 = mul(ring, b3);
    b4 = mul(ring, b4);
    root[3] = ring;
    const V3<T> n3 = cross(b4, b3);
    n2 = cross(b3, b2);
    const T a32 = signed_angle(n3, n2, b3);
    const M3<T> pinky = rotation_matrix(a32 + canon[2] + pra[2], b3);
    root[4] = mul(pinky, ring);
    for (int i = 0; i < N; ++i) {   // N = 20: all bones (the kinematic chains inherit their root's matrix); 5: the roots alone
        mat[i] = root[i % 5];
        out[i] = mul(mat[i], bones[i]);
    }
}
// normalize_root_bone_angles (:1033-1107); pra entries 3..6
template <typename S, int N> HN_PC_FN void normalize_root_bone_angles(const V3<Dual<S>> (&bones)[N], const Dual<S> (&pra)[N_PRA], V3<Dual<S>> (&out)[N], M3<Dual<S>> (&mat)[N]) {
    using T = Dual<S>;
    const S canon[4] = {S(0.4), S(0.2), S(0.2), S(0.2)};
    V3<T> b0 = bones[0], b1 = bones[1], b2 = bones[2], b3 = bones[3], b4 = bones[4];
    M3<T> root[5];
    for (int i = 0; i < 5; ++i) root[i] = eye3<T>();
    const V3<T> n1 = unit(cross(b2, b1), S(1e-8));
    const T f21 = signed_angle(b2, b1, n1);
    const M3<T> index_t = rotation_matrix(canon[1] - f21 + pra[3], n1);
    root[1] = index_t;
    b1 = mul(index_t, b1);
    b0 = mul(index_t, b0);
    const V3<T> n0 = unit(cross(b1, b0), S(1e-8));
    const T f10 = signed_angle(b1, b0, n0);
    const M3<T> thumb_t = rotation_matrix(canon[0] - f10 + pra[4], n0);
    root[0] = mul(thumb_t, index_t);
    const V3<T> n2 = unit(cross(b3, b2), S(1e-8));
    const T f32 = signed_angle(b3, b2, n2);
    const M3<T> ring_t = rotation_matrix(f32 - canon[2] + pra[5], n2);
    root[3] = ring_t;
    b3 = mul(ring_t, b3);
    b4 = mul(ring_t, b4);
    const V3<T> n3 = unit(cross(b4, b3), S(1e-8));
    const T f43 = signed_angle(b4, b3, n3);
    const M3<T> pinky_t = rotation_matrix(f43 - canon[3] + pra[6], n3);
    root[4] = mul(pinky_t, ring_t);
    for (int i = 0; i < N; ++i) {   // N = 20: all bones (the kinematic chains inherit their root's matrix); 5: the roots alone
        mat[i] = root[i % 5];
        out[i] = mul(mat[i], bones[i]);
    }
}

// the flexion / abduction angle pair of a bone in a local frame (compute_rot_angles :731-766; the same construction
// with the other sign convention of the second angle inside compute_local_coordinate_system :664-690)
template <typename S> HN_PC_FN void bone_angles(const V3<Dual<S>>& lc, bool abduction_sign_of_rot_angles, Dual<S>& a_xz, Dual<S>& a_yz) {
    using T = Dual<S>;
    const V3<T> proj = {{lc.x[0], T(S(0)), lc.x[2]}};
    const T nrm = max_c(norm(proj), S(1e-9));
    T d_xz = proj.x[2];
    if (fabs((double)d_xz.v) < 1e-6) d_xz = T(S(0));
    d_xz = clip_c(d_xz / nrm, S(-1) + S(1e-6), S(1) - S(1e-6));
    a_xz = acos_d(d_xz);
    if (proj.x[0].v + S(1e-6) < S(0)) a_xz = -a_xz;
    const T d_yz = clip_c(dot(proj, lc) / nrm, S(-1) + S(1e-6), S(1) - S(1e-6));
    a_yz = acos_d(d_yz);
    const bool neg = abduction_sign_of_rot_angles ? (lc.x[1].v + S(1e-6) > S(0)) : (lc.x[1].v + S(1e-6) < S(0));
    if (neg) a_yz = -a_yz;
}

// compute_local_coordinate_system (:596-722): returned detached, so values only (the tangents are dropped on entry)
template <typename S> HN_PC_FN void local_coordinate_system(const V3<Dual<S>> (&bones_in)[20], M3<Dual<S>> (&cs)[20]) {
    using T = Dual<S>;
    V3<T> bones[20];
    for (int i = 0; i < 20; ++i) bones[i] = detach3(bones_in[i]);
    V3<T> pn[4];
    for (int k = 0; k < 4; ++k) pn[k] = unit(cross(bones[k], bones[k + 1]), S(1e-9));
    V3<T> fpn[5];
    fpn[0] = pn[0];
    fpn[1] = pn[1];
    fpn[2] = scale(pn[1] + pn[2], T(S(0.5)));
    fpn[3] = scale(pn[2] + pn[3], T(S(0.5)));
    fpn[4] = pn[3];
    V3<T> x[5], y[5], z[5];
    for (int k = 0; k < 5; ++k) {
        cs[k] = eye3<T>();
        z[k] = bones[k];
        y[k] = cross(bones[k], fpn[k]);
        x[k] = cross(y[k], z[k]);
        x[k] = unit(x[k], S(1e-9));
        y[k] = unit(y[k], S(1e-9));
        for (int c = 0; c < 3; ++c) {
            cs[5 + k].m[0][c] = x[k].x[c];
            cs[5 + k].m[1][c] = y[k].x[c];
            cs[5 + k].m[2][c] = z[k].x[c];
        }
    }
    const V3<T> y_axis = {{T(S(0)), T(S(1)), T(S(0))}}, x_axis = {{T(S(1)), T(S(0)), T(S(0))}};
    for (int lev = 2; lev < 4; ++lev)
        for (int k = 0; k < 5; ++k) {
            const int idx = 5 * lev + k;
            const M3<T>& pc = cs[idx - 5];
            const V3<T> lbv2 = mul(pc, bones[idx - 5]);
            T a_xz, a_yz;
            bone_angles(lbv2, false, a_xz, a_yz);
            const M3<T> pct = transpose(pc);
            const SinCos<S> sc_xz = sincos_d(a_xz), sc_yz = sincos_d(-a_yz);
            const V3<T> axis_xz = mul(pct, y_axis);
            const V3<T> axis_y = mul(pct, rotate(x_axis, y_axis, sc_xz));
            if (!(fabs((double)a_xz.v) < 1e-6)) {
                x[k] = rotate(x[k], axis_xz, sc_xz);
                y[k] = rotate(y[k], axis_xz, sc_xz);
                z[k] = rotate(z[k], axis_xz, sc_xz);
            }
            if (!(fabs((double)a_yz.v) < 1e-6)) {
                x[k] = rotate(x[k], axis_y, sc_yz);
                y[k] = rotate(y[k], axis_y, sc_yz);
                z[k] = rotate(z[k], axis_y, sc_yz);
            }
            for (int c = 0; c < 3; ++c) {
                cs[idx].m[0][c] = x[k].x[c];
                cs[idx].m[1][c] = y[k].x[c];
                cs[idx].m[2][c] = z[k].x[c];
            }
        }
    for (int i = 0; i < 20; ++i) cs[i] = detachM(cs[i]);
}

// compute_rotation_matrix (:811-875) with the straight-hand canonical angles (all zero, :411-432)
template <typename S> HN_PC_FN void rotation_matrices(const Dual<S> (&a_xz)[20], const Dual<S> (&a_yz)[20], const Dual<S>* jra /* [20] or null */, M3<Dual<S>> (&r)[20]) {
    using T = Dual<S>;
    const V3<T> x = {{T(S(1)), T(S(0)), T(S(0))}}, y = {{T(S(0)), T(S(1)), T(S(0))}};
    for (int i = 0; i < 20; ++i) {
        if (i < 5) {
            r[i] = eye3<T>();   // the root bones' rotation is masked (:857)
            continue;
        }
        const V3<T> rotated_x = rotate(x, y, a_xz[i]);
        T abduction = -a_yz[i];
        if (jra != nullptr && i < 10) abduction = abduction + jra[i - 5];
        const M3<T> r1 = rotation_matrix(abduction, rotated_x);
        T flexion = -a_xz[i];
        if (jra != nullptr) flexion = flexion + jra[i];   // joint_refine_angle[:, 5:] onto bones 5..19
        const M3<T> r2 = rotation_matrix(flexion, y);
        r[i] = mul(r2, r1);
    }
}

// the part shared by PoseConverter.forward and get_refine_3d_joint: canonical-frame joints -> normalised bones, frames,
// the per-bone matrix `trans_mat_without_scale_translation` (tm) and the root normalisation (rbn)
template <typename S>
HN_PC_FN void converter_core(const V3<Dual<S>> (&joints_in)[21], bool is_right, const Dual<S>* jra, const Dual<S> (&pra)[N_PRA], V3<Dual<S>> (&bones)[20],
                             Dual<S> (&bl)[20], V3<Dual<S>> (&kpar)[20], M3<Dual<S>> (&rbn)[20], M3<Dual<S>> (&tm)[20]) {
    using T = Dual<S>;
    // preprocess_joints (:769-808): shift_factor = 0; left hands mirrored in x
    V3<T> joints[21];
    for (int i = 0; i < 21; ++i) {
        joints[i] = joints_in[i];
        if (!is_right) joints[i].x[0] = -joints[i].x[0];
    }
    V3<T> b0[20];
    kp_to_bones(joints, b0, bl, kpar);
    V3<T> b1[20];
    M3<T> plane_mat[20], angle_mat[20];
    normalize_root_planes(b0, pra, b1, plane_mat);
    normalize_root_bone_angles(b1, pra, bones, angle_mat);
    for (int i = 0; i < 20; ++i) rbn[i] = mul(angle_mat[i], plane_mat[i]);
    M3<T> cs[20];
    local_coordinate_system(bones, cs);
    T a_xz[20], a_yz[20];
    for (int i = 0; i < 20; ++i) bone_angles(mul(cs[i], bones[i]), true, a_xz[i], a_yz[i]);
    M3<T> rot[20];
    rotation_matrices(a_xz, a_yz, jra, rot);
    // compute_adjusted_transpose (:939-962) and trans_mat_without_scale_translation
    for (int i = 0; i < 20; ++i) {
        M3<T> lct = transpose(cs[i]);
        if (i >= 10 && i < 15) lct = mul(lct, rot[i - 5]);
        if (i >= 15) lct = mul(lct, mul(rot[i - 5], rot[i - 10]));
        tm[i] = mul(lct, mul(rot[i], cs[i]));
    }
}

// PoseConverter.forward (:1109-1179): canonical-frame joints (biomech order) -> trans_mat [21] as affine maps (R | t); the
// fourth row is (0, 0, 0, 1).  trans_mat = bone_to_kp . tm . root_norm . kp_to_bone (:1157-1162) with
//   kp_to_bone = scale(1 / l) translate(-parent),  bone_to_kp = translate(tr) scale(l)   (compute_bone_to_kp_mat :564-594,
//   tr = the unposed bones accumulated along the finger):   R = l (tm rbn) / l = tm rbn,   t = tr - R parent
// -- the four 4 x 4 products of the reference collapse to one 3 x 3 product and one matrix-vector product per bone.
template <typename S> HN_PC_FN void converter_forward(const V3<Dual<S>> (&joints)[21], bool is_right, M3<Dual<S>> (&outR)[21], V3<Dual<S>> (&outT)[21]) {
    using T = Dual<S>;
    V3<T> bones[20], kpar[20];
    T bl[20], pra[N_PRA];
    M3<T> rbn[20], tm[20];
    converter_core(joints, is_right, (const T*)nullptr, pra, bones, bl, kpar, rbn, tm);
    V3<T> lc[20];   // local_coords_after_unpose, times the bone length
    for (int i = 0; i < 20; ++i) lc[i] = scale(mul(tm[i], bones[i]), bl[i]);
    outR[0] = eye3<T>();
    outT[0] = {{T(S(0)), T(S(0)), T(S(0))}};
    for (int i = 0; i < 20; ++i) {
        V3<T> tr = {{T(S(0)), T(S(0)), T(S(0))}};
        for (int p = i - 5; p >= 0; p -= 5) tr = tr + lc[p];
        outR[i + 1] = mul(tm[i], rbn[i]);
        outT[i + 1] = tr - mul(outR[i + 1], kpar[i]);
    }
}

// PoseConverter.get_refine_3d_joint (:1183-1229): canonical-frame joints + refine angles -> joints [21][3] of the refined pose
// (finger-major order, as forward_get_3djoint_use_bone_and_bone_length :261-277 builds it)
template <typename S>
HN_PC_FN void refine_3d_joint(const V3<Dual<S>> (&joints)[21], bool is_right, const S (&mean_bl)[20], const Dual<S> (&jra)[N_JRA], const Dual<S> (&pra)[N_PRA],
                              V3<Dual<S>> (&out)[21]) {
    using T = Dual<S>;
    // self.initial_bone_vec (:434-453): a data table of the reference (unit bone directions of its rest pose)
    const S ibv[20][3] = {{4.4889e-01, -8.4880e-01, -2.7935e-01}, {1.9867e-01, -9.8007e-01, 0.0000e+00},  {2.0004e-07, -1.0000e+00, 0.0000e+00},
                          {-1.9471e-01, -9.8007e-01, -3.9469e-02}, {-3.7001e-01, -9.2185e-01, -1.1528e-01}, {4.4889e-01, -8.4880e-01, -2.7935e-01},
                          {1.9867e-01, -9.8007e-01, 1.1921e-07},   {2.8685e-07, -1.0000e+00, 0.0000e+00},  {-1.9471e-01, -9.8007e-01, -3.9470e-02},
                          {-3.7001e-01, -9.2185e-01, -1.1528e-01}, {4.4889e-01, -8.4880e-01, -2.7935e-01}, {1.9867e-01, -9.8007e-01, 1.4901e-07},
                          {1.9870e-06, -1.0000e+00, 2.3842e-07},   {-1.9471e-01, -9.8007e-01, -3.9470e-02}, {-3.7001e-01, -9.2185e-01, -1.1528e-01},
                          {4.4889e-01, -8.4880e-01, -2.7935e-01},  {1.9867e-01, -9.8007e-01, 8.9407e-08},  {-3.4117e-06, -1.0000e+00, -2.1979e-07},
                          {-1.9471e-01, -9.8007e-01, -3.9469e-02}, {-3.7001e-01, -9.2185e-01, -1.1528e-01}};
    V3<T> bones[20], kpar[20];
    T bl[20];
    M3<T> rbn[20], tm[20];
    converter_core(joints, is_right, jra, pra, bones, bl, kpar, rbn, tm);
    V3<T> p_bone[20];
    for (int i = 0; i < 20; ++i) {
        const M3<T> inv = inverse(mul(tm[i], rbn[i]));
        const V3<T> v = {{T(ibv[i][0]), T(ibv[i][1]), T(ibv[i][2])}};
        p_bone[i] = mul(inv, v);
    }
    out[0] = {{T(S(0)), T(S(0)), T(S(0))}};
    int n = 1;
    for (int k = 0; k < 5; ++k) {
        V3<T> start = {{T(S(0)), T(S(0)), T(S(0))}};
        for (int i = 0; i < 4; ++i) {
            const int idx = 5 * i + k;
            start = start + scale(p_bone[idx], T(mean_bl[idx]));
            out[n++] = start;
        }
    }
}

// rot6d_to_matrix (utils/utils.py:11-30): rot_6d viewed [3][2]
template <typename S> HN_PC_FN M3<Dual<S>> rot6d_to_matrix(const Dual<S> (&r6)[6]) {
    using T = Dual<S>;
    const V3<T> a1 = {{r6[0], r6[2], r6[4]}}, a2 = {{r6[1], r6[3], r6[5]}};
    const V3<T> b1 = unit(a1, S(1e-12));
    const V3<T> b2 = unit(a2 - scale(b1, dot(b1, a2)), S(1e-12));
    const V3<T> b3 = cross(b1, b2);
    M3<T> R;
    for (int i = 0; i < 3; ++i) {
        R.m[i][0] = b1.x[i];
        R.m[i][1] = b2.x[i];
        R.m[i][2] = b3.x[i];
    }
    return R;
}

// The chain of fitting_single.py:206-226.  ori_pose: the predicted joints (MANO order, fixed); mean_bl: cur_bone_length
// (20, biomech bone order, fixed); in: the N_IN differentiable inputs [joint_refine_angle 20 | palm_refine_angle 7 |
// palm_rot_refine 6 (row-major [3][2]) | palm_trans_refine 3] as duals.  out: bone_transformation_inv [21][16] then
// joint_3d [21][3] (MANO order), N_OUT duals.
template <typename S>
HN_PC_FN void pose_chain(const S (&ori_pose)[21][3], const S (&mean_bl)[20], bool is_right, const Dual<S> (&in)[N_IN], Dual<S> (&out)[N_OUT]) {
    using T = Dual<S>;
    T jra[N_JRA], pra[N_PRA], r6[6];
    for (int i = 0; i < N_JRA; ++i) jra[i] = in[i];
    for (int i = 0; i < N_PRA; ++i) pra[i] = in[N_JRA + i] * S(0.1);   // palm_refine_angle * 0.1 (fitting_single.py:210)
    for (int i = 0; i < 6; ++i) r6[i] = in[N_JRA + N_PRA + i];
    const V3<T> t_palm = {{in[N_JRA + N_PRA + 6], in[N_JRA + N_PRA + 7], in[N_JRA + N_PRA + 8]}};
    // :206-208  predicted joints -> biomech order -> canonical frame
    V3<T> kps[21];
    for (int i = 0; i < 21; ++i) {
        const int m = mano_to_biomech(i);
        kps[i] = {{T(ori_pose[m][0]), T(ori_pose[m][1]), T(ori_pose[m][2])}};
    }
    M3<T> R1;
    V3<T> t1;
    canonical_transform(kps, is_right, R1, t1);
    V3<T> pal[21];
    for (int i = 0; i < 21; ++i) pal[i] = mul(R1, kps[i]) + t1;
    // :209-210
    V3<T> j3[21];
    refine_3d_joint(pal, is_right, mean_bl, jra, pra, j3);
    // :211-212  back through the inverse of the canonical transform
    const M3<T> R1i = inverse(R1);
    const V3<T> t1i = scale(mul(R1i, t1), T(S(-1)));
    for (int i = 0; i < 21; ++i) j3[i] = mul(R1i, j3[i]) + t1i;
    // :213-217  palm rotation about the root joint, palm translation
    const M3<T> Rp = rot6d_to_matrix(r6);
    const V3<T> root = j3[0];
    for (int i = 0; i < 21; ++i) j3[i] = mul(Rp, j3[i] - root) + root + t_palm;
    // :218-222  refined joints -> biomech order -> canonical frame (the transform itself is detached) -> PoseConverter
    V3<T> kps2[21];
    for (int i = 0; i < 21; ++i) kps2[i] = j3[mano_to_biomech(i)];
    M3<T> R2;
    V3<T> t2;
    canonical_transform(kps2, is_right, R2, t2);
    V3<T> pal2[21];
    for (int i = 0; i < 21; ++i) pal2[i] = mul(R2, kps2[i]) + t2;
    M3<T> tR[21];
    V3<T> tT[21];
    converter_forward(pal2, is_right, tR, tT);
    // :223-226  back to MANO order, times the canonical transform (R2 | t2)
    for (int i = 0; i < 21; ++i) {
        const int b = biomech_to_mano(i);
        const M3<T> R = mul(tR[b], R2);
        const V3<T> t = mul(tR[b], t2) + tT[b];
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) out[16 * i + 4 * r + c] = R.m[r][c];
            out[16 * i + 4 * r + 3] = t.x[r];
        }
        for (int c = 0; c < 4; ++c) out[16 * i + 12 + c] = T(S(c == 3 ? 1 : 0));
    }
    for (int i = 0; i < 21; ++i)
        for (int c = 0; c < 3; ++c) out[21 * 16 + 3 * i + c] = j3[i].x[c];
}

// ---- the same chain, one FINGER at a time --------------------------------------------------------------------------------
// Everything above bone level is per finger (bone i = 5 L + f: level L of finger f; its joint is biomech joint i + 1 =
// MANO joint 1 + 4 f + L); only three things cross fingers: the five raw root bones (twice: the root normalisations and the
// finger planes of each PoseConverter pass need all of them) and the level-0 joints of fingers 1 and 2 (the second
// canonical transform).  So the chain splits into four phases per (input direction, finger) with three exchanges between
// them: the device kernel runs one THREAD per (direction, finger) -- a fifth of the serial work, arrays of 4 instead of 20
// -- and the host oracle runs the phases in loops.  Same arithmetic, statement for statement, as pose_chain() above (the
// host oracle checks the two against each other).
template <typename T> struct FingerState {
    V3<T> raw[4], kpar[4], bones[4];   // raw unit bones, parent joints, normalised bones
    T bl[4];
    M3<T> rbn, tm[4];                  // the finger's root normalisation; per bone trans_mat_without_scale_translation
    M3<T> Rc;                          // canonical transform in use (R | t)
    V3<T> tc;
    V3<T> J[4], J0;                    // phase B: the finger's refined joints and the root joint (world frame)
};
// own joints (canonical frame) -> raw bones; returns the finger's raw root bone
template <typename S> HN_PC_FN V3<Dual<S>> finger_bones(int f, const V3<Dual<S>>& root, const V3<Dual<S>> (&jt)[4], bool is_right, FingerState<Dual<S>>& st) {
    using T = Dual<S>;
    V3<T> r0 = root, j[4];
    for (int L = 0; L < 4; ++L) j[L] = jt[L];
    if (!is_right) {
        r0.x[0] = -r0.x[0];
        for (int L = 0; L < 4; ++L) j[L].x[0] = -j[L].x[0];
    }
    for (int L = 0; L < 4; ++L) {
        const V3<T> par = L == 0 ? r0 : j[L - 1];
        const V3<T> b = j[L] - par;
        st.bl[L] = max_c(norm(b), S(1e-9));
        st.raw[L] = scale(b, T(S(1)) / st.bl[L]);
        st.kpar[L] = par;
    }
    (void)f;
    return st.raw[0];
}
// all five raw root bones -> this finger's normalised bones, frames and tm (converter_core from normalize_root_planes on)
template <typename S> __attribute__((always_inline)) HN_PC_FN void finger_core(int f, const V3<Dual<S>> (&RB)[5], const Dual<S>* jra, const Dual<S> (&pra)[N_PRA], FingerState<Dual<S>>& st) {
    using T = Dual<S>;
    // the two root normalisations on the root bones alone (their matrices are per finger; levels 1..3 inherit them)
    V3<T> o1[5], o2[5];
    M3<T> pm[5], am[5];
    normalize_root_planes(RB, pra, o1, pm);
    normalize_root_bone_angles(o1, pra, o2, am);
    st.rbn = mul(am[f], pm[f]);
    for (int L = 0; L < 4; ++L) st.bones[L] = mul(am[f], mul(pm[f], st.raw[L]));
    // compute_local_coordinate_system, this finger's chain (detached)
    V3<T> nb[5];
    for (int k = 0; k < 5; ++k) nb[k] = detach3(o2[k]);
    V3<T> pn[4];
    for (int k = 0; k < 4; ++k) pn[k] = unit(cross(nb[k], nb[k + 1]), S(1e-9));
    const V3<T> fpn = f == 0 ? pn[0] : (f == 1 ? pn[1] : (f == 2 ? scale(pn[1] + pn[2], T(S(0.5))) : (f == 3 ? scale(pn[2] + pn[3], T(S(0.5))) : pn[3])));
    V3<T> bd[4];
    for (int L = 0; L < 4; ++L) bd[L] = detach3(st.bones[L]);
    M3<T> cs[4];
    cs[0] = eye3<T>();
    V3<T> z = bd[0], y = cross(bd[0], fpn), x = cross(y, z);
    x = unit(x, S(1e-9));
    y = unit(y, S(1e-9));
    for (int c = 0; c < 3; ++c) {
        cs[1].m[0][c] = x.x[c];
        cs[1].m[1][c] = y.x[c];
        cs[1].m[2][c] = z.x[c];
    }
    const V3<T> y_axis = {{T(S(0)), T(S(1)), T(S(0))}}, x_axis = {{T(S(1)), T(S(0)), T(S(0))}};
    for (int L = 2; L < 4; ++L) {
        const M3<T>& pc = cs[L - 1];
        const V3<T> lbv2 = mul(pc, bd[L - 1]);
        T a_xz, a_yz;
        bone_angles(lbv2, false, a_xz, a_yz);
        const M3<T> pct = transpose(pc);
        const SinCos<S> sc_xz = sincos_d(a_xz), sc_yz = sincos_d(-a_yz);
        const V3<T> axis_xz = mul(pct, y_axis);
        const V3<T> axis_y = mul(pct, rotate(x_axis, y_axis, sc_xz));
        if (!(fabs((double)a_xz.v) < 1e-6)) {
            x = rotate(x, axis_xz, sc_xz);
            y = rotate(y, axis_xz, sc_xz);
            z = rotate(z, axis_xz, sc_xz);
        }
        if (!(fabs((double)a_yz.v) < 1e-6)) {
            x = rotate(x, axis_y, sc_yz);
            y = rotate(y, axis_y, sc_yz);
            z = rotate(z, axis_y, sc_yz);
        }
        for (int c = 0; c < 3; ++c) {
            cs[L].m[0][c] = x.x[c];
            cs[L].m[1][c] = y.x[c];
            cs[L].m[2][c] = z.x[c];
        }
    }
    for (int L = 0; L < 4; ++L) cs[L] = detachM(cs[L]);
    // angles, rotation matrices (compute_rotation_matrix), adjusted transpose, tm
    M3<T> rot[4];
    const V3<T> ex = {{T(S(1)), T(S(0)), T(S(0))}}, ey = {{T(S(0)), T(S(1)), T(S(0))}};
    rot[0] = eye3<T>();
    for (int L = 1; L < 4; ++L) {
        T a_xz, a_yz;
        bone_angles(mul(cs[L], st.bones[L]), true, a_xz, a_yz);
        const int i = 5 * L + f;
        const V3<T> rotated_x = rotate(ex, ey, a_xz);
        T abduction = -a_yz;
        if (jra != nullptr && i < 10) abduction = abduction + jra[i - 5];
        const M3<T> r1 = rotation_matrix(abduction, rotated_x);
        T flexion = -a_xz;
        if (jra != nullptr) flexion = flexion + jra[i];
        rot[L] = mul(rotation_matrix(flexion, ey), r1);
    }
    for (int L = 0; L < 4; ++L) {
        M3<T> lct = transpose(cs[L]);
        if (L == 2) lct = mul(lct, rot[1]);
        if (L == 3) lct = mul(lct, mul(rot[2], rot[1]));
        st.tm[L] = mul(lct, mul(rot[L], cs[L]));
    }
}
// inputs of one (direction) evaluation, shared by the phases
template <typename S> struct ChainIn {
    const S (*ori_pose)[3];
    const S* mean_bl;
    bool is_right;
    Dual<S> jra[N_JRA], pra[N_PRA], r6[6];
    V3<Dual<S>> t_palm;
};
template <typename S> HN_PC_FN void chain_inputs(const S (&ori_pose)[21][3], const S (&mean_bl)[20], bool is_right, const Dual<S> (&in)[N_IN], ChainIn<S>& ci) {
    ci.ori_pose = ori_pose;
    ci.mean_bl = mean_bl;
    ci.is_right = is_right;
    for (int i = 0; i < N_JRA; ++i) ci.jra[i] = in[i];
    for (int i = 0; i < N_PRA; ++i) ci.pra[i] = in[N_JRA + i] * S(0.1);
    for (int i = 0; i < 6; ++i) ci.r6[i] = in[N_JRA + N_PRA + i];
    ci.t_palm = {{in[N_JRA + N_PRA + 6], in[N_JRA + N_PRA + 7], in[N_JRA + N_PRA + 8]}};
}
// phase A: predicted joints -> canonical frame -> this finger's raw bones; returns its raw root bone
template <typename S> HN_PC_FN V3<Dual<S>> chain_phase_a(int f, const ChainIn<S>& ci, FingerState<Dual<S>>& st) {
    using T = Dual<S>;
    V3<T> kps[21];   // (values only: the predicted joints are data)
    for (int i = 0; i < 21; ++i) {
        const int m = mano_to_biomech(i);
        kps[i] = {{T(ci.ori_pose[m][0]), T(ci.ori_pose[m][1]), T(ci.ori_pose[m][2])}};
    }
    canonical_transform(kps, ci.is_right, st.Rc, st.tc);
    V3<T> jt[4];
    for (int L = 0; L < 4; ++L) jt[L] = mul(st.Rc, kps[5 * L + f + 1]) + st.tc;
    return finger_bones(f, mul(st.Rc, kps[0]) + st.tc, jt, ci.is_right, st);
}
// phase B: refine (get_refine_3d_joint), back to the world frame, palm motion; returns the finger's level-0 joint
template <typename S> HN_PC_FN V3<Dual<S>> chain_phase_b(int f, const ChainIn<S>& ci, const V3<Dual<S>> (&RB)[5], FingerState<Dual<S>>& st) {
    using T = Dual<S>;
    const S ibv[20][3] = {{4.4889e-01, -8.4880e-01, -2.7935e-01}, {1.9867e-01, -9.8007e-01, 0.0000e+00},  {2.0004e-07, -1.0000e+00, 0.0000e+00},
                          {-1.9471e-01, -9.8007e-01, -3.9469e-02}, {-3.7001e-01, -9.2185e-01, -1.1528e-01}, {4.4889e-01, -8.4880e-01, -2.7935e-01},
                          {1.9867e-01, -9.8007e-01, 1.1921e-07},   {2.8685e-07, -1.0000e+00, 0.0000e+00},  {-1.9471e-01, -9.8007e-01, -3.9470e-02},
                          {-3.7001e-01, -9.2185e-01, -1.1528e-01}, {4.4889e-01, -8.4880e-01, -2.7935e-01}, {1.9867e-01, -9.8007e-01, 1.4901e-07},
                          {1.9870e-06, -1.0000e+00, 2.3842e-07},   {-1.9471e-01, -9.8007e-01, -3.9470e-02}, {-3.7001e-01, -9.2185e-01, -1.1528e-01},
                          {4.4889e-01, -8.4880e-01, -2.7935e-01},  {1.9867e-01, -9.8007e-01, 8.9407e-08},  {-3.4117e-06, -1.0000e+00, -2.1979e-07},
                          {-1.9471e-01, -9.8007e-01, -3.9469e-02}, {-3.7001e-01, -9.2185e-01, -1.1528e-01}};
    finger_core(f, RB, ci.jra, ci.pra, st);
    const M3<T> R1i = inverse(st.Rc);
    const V3<T> t1i = scale(mul(R1i, st.tc), T(S(-1)));
    const M3<T> Rp = rot6d_to_matrix(ci.r6);
    const V3<T> root = t1i;   // the refined pose's root joint is the canonical origin
    V3<T> start = {{T(S(0)), T(S(0)), T(S(0))}};
    for (int L = 0; L < 4; ++L) {
        const int i = 5 * L + f;
        const M3<T> inv = inverse(mul(st.tm[L], st.rbn));
        const V3<T> v = {{T(ibv[i][0]), T(ibv[i][1]), T(ibv[i][2])}};
        start = start + scale(mul(inv, v), T(ci.mean_bl[i]));
        const V3<T> w = mul(R1i, start) + t1i;                     // back through the inverse canonical transform
        st.J[L] = mul(Rp, w - root) + root + ci.t_palm;             // palm rotation about the root joint + translation
    }
    st.J0 = root + ci.t_palm;
    return st.J[0];
}
// phase C: the refined joints -> second canonical frame -> raw bones; J1, J2: the level-0 joints of fingers 1 and 2
template <typename S> HN_PC_FN V3<Dual<S>> chain_phase_c(int f, const ChainIn<S>& ci, const V3<Dual<S>>& J1, const V3<Dual<S>>& J2, FingerState<Dual<S>>& st) {
    using T = Dual<S>;
    V3<T> kp[21];
    for (int i = 0; i < 21; ++i) kp[i] = {{T(S(0)), T(S(0)), T(S(0))}};
    kp[0] = st.J0;   // canonical_transform reads joints 0, 2, 3 (biomech): the root and the level-0 joints of fingers 1, 2
    kp[2] = J1;
    kp[3] = J2;
    canonical_transform(kp, ci.is_right, st.Rc, st.tc);
    V3<T> jt[4];
    for (int L = 0; L < 4; ++L) jt[L] = mul(st.Rc, st.J[L]) + st.tc;
    return finger_bones(f, mul(st.Rc, st.J0) + st.tc, jt, ci.is_right, st);
}
// phase D: PoseConverter.forward for this finger -> its rows of bone_transformation_inv and joint_3d.  put(index, value)
// receives output `index` of the N_OUT (finger 0 also writes the root's rows).
template <typename S, typename Put> HN_PC_FN void chain_phase_d(int f, const ChainIn<S>& ci, const V3<Dual<S>> (&RB)[5], FingerState<Dual<S>>& st, Put&& put) {
    using T = Dual<S>;
    T pra0[N_PRA];
    finger_core(f, RB, (const T*)nullptr, pra0, st);
    auto put_affine = [&](int im, const M3<T>& R, const V3<T>& t) {
        const M3<T> RR = mul(R, st.Rc);
        const V3<T> tt = mul(R, st.tc) + t;
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) put(16 * im + 4 * r + c, RR.m[r][c]);
            put(16 * im + 4 * r + 3, tt.x[r]);
        }
        for (int c = 0; c < 4; ++c) put(16 * im + 12 + c, T(S(c == 3 ? 1 : 0)));
    };
    V3<T> tr = {{T(S(0)), T(S(0)), T(S(0))}};
    for (int L = 0; L < 4; ++L) {
        const M3<T> R = mul(st.tm[L], st.rbn);
        put_affine(1 + 4 * f + L, R, tr - mul(R, st.kpar[L]));
        tr = tr + scale(mul(st.tm[L], st.bones[L]), st.bl[L]);
        for (int c = 0; c < 3; ++c) put(21 * 16 + 3 * (1 + 4 * f + L) + c, st.J[L].x[c]);
    }
    if (f == 0) {
        put_affine(0, eye3<T>(), V3<T>{{T(S(0)), T(S(0)), T(S(0))}});
        for (int c = 0; c < 3; ++c) put(21 * 16 + c, st.J0.x[c]);
    }
    (void)ci;
}

// the four phases in loops over the fingers: what the device kernel does with one thread per finger and three exchanges
template <typename S>
HN_PC_FN void pose_chain_by_finger(const S (&ori_pose)[21][3], const S (&mean_bl)[20], bool is_right, const Dual<S> (&in)[N_IN], Dual<S> (&out)[N_OUT]) {
    using T = Dual<S>;
    ChainIn<S> ci;
    chain_inputs(ori_pose, mean_bl, is_right, in, ci);
    FingerState<T> st[5];
    V3<T> RB[5], J[5];
    for (int f = 0; f < 5; ++f) RB[f] = chain_phase_a(f, ci, st[f]);
    for (int f = 0; f < 5; ++f) J[f] = chain_phase_b(f, ci, RB, st[f]);
    for (int f = 0; f < 5; ++f) RB[f] = chain_phase_c(f, ci, J[1], J[2], st[f]);
    for (int f = 0; f < 5; ++f) chain_phase_d(f, ci, RB, st[f], [&](int idx, const T& v) { out[idx] = v; });
}

}  // namespace pose
}  // namespace hn
