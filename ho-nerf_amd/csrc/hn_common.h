// Shared host/device definitions of the honerf library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/honerf.h"

namespace hn {

// ---- error plumbing ----------------------------------------------------------------------
void set_error(const char* fmt, ...);
#define HN_CHECK_HIP(expr)                                                                    \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            hn::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return HN_EHIP;                                                                   \
        }                                                                                     \
    } while (0)
#define HN_REQUIRE(cond, ...)                  \
    do {                                       \
        if (!(cond)) {                         \
            hn::set_error(__VA_ARGS__);        \
            return HN_EINVAL;                  \
        }                                      \
    } while (0)
#define HN_LAUNCH_CHECK() HN_CHECK_HIP(hipGetLastError())
#define HN_TRY_RC(expr)                  \
    do {                                 \
        const int _rc = (expr);          \
        if (_rc != HN_OK) return _rc;    \
    } while (0)

// Per-device state (one process may drive several GPUs, from several threads).
int current_device();          // hipGetDevice, -1 on failure
int device_cus();              // CU count of the CURRENT device (cached per device; 0 if none)
// Device-side sample count of the NEXT hand-field launches of this host thread (hn_api.hip sets it around the launches over a
// compacted sample list and clears it again); NULL otherwise.
const int* launch_n_pts_dev();
void set_launch_n_pts_dev(const int* p);
const int* launch_orig_idx();
void set_launch_orig_idx(const int* p);
// The frame table of a FRAME-ALIGNED compact list (hn_api.hip, k_hand_compact_write: [first slot per frame, af + 1 | live samples per
// frame, af]; n_pts_dev[2] = af) for the next hand-field ADJOINT launch of this host thread: k_pose_part_reduce takes a frame's rows
// from it.  NULL: a dense list, or the plain compact layout of a one-frame launch.
const int* launch_frame_seg();
void set_launch_frame_seg(const int* p);
bool launch_dir_per_sample();     // hn_api.hip: the object adjoint launch writes d loss / d rays_d per sample (set around the launch)
int quad_max_blocks_override();   // hn_debug_quad_max_blocks: -1 = default selection of the latency-form kernels
int pace_phantom_members();    // hn_debug_pace_phantom: members that never arrive at the XCD meetings (timeout-path test hook), 0 = off
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device): `mask` is the kernel's own
// bit set of devices already configured (a static std::atomic<uint64_t> next to the launch).
int ensure_dynamic_lds(const void* kernel, int bytes, void* mask_atomic_u64);
// hn_pose_chain.hip
int pose_chain(const float* ori_pose, const float* bone_len, const unsigned char* is_right, const float* in, int n_frames, float* bt_inv,
               float* joint_3d, float* jac, hipStream_t s);
int pose_chain_bwd(const float* jac, const float* g_bt_inv, const float* g_joint_3d, int n_frames, float* g_in, hipStream_t s);
// hn_pose_rigid.hip
int rigid_pose(const float* bt_inv0, const float* joints0, const float* Ro_pred, const float* To_pred, const float* params, int n_frames, int with_palm,
               float* out, float* jac, hipStream_t s);
int verts_loss(const float* Ra, const float* ta, const float* Rb, const float* tb, const float* verts, int n_verts, int n_pairs, float* loss, float* gR,
               float* gt, hipStream_t s);
int jacobian_vjp(const float* jac, const float* g, int n_frames, int n_out, int n_in, float* out, hipStream_t s);
// what the wave form of up_sample can do in front of the round itself (hn_sampling.hip, UpsExtra): the previous round's cat_z_vals
// (m_prev > 0: merge zp / sp [n_rays, m_prev] into the old rows, merged rows -> z_out / sdf_out) or the gather of the hand's coarse
// sdf row through a compaction record (pos, n_dev, sdf_c; dense row -> sdf_out)
struct UpsPre {
    const float *zp, *sp;
    int m_prev, quirk_p;
    float *z_out, *sdf_out;
    const int *pos, *n_dev;
    const float* sdf_c;
};
int leaf_rows_gather(const float* const* leaves6, const long long* rows, int F, int n, float* prm_h, float* prm_o, hipStream_t s);
int leaf_rows_scatter(const float* g, const long long* rows, int F, int n, float* out, hipStream_t s);
int pose_side_vjp(const float* jac_h, const float* jac_o, const float* g_bt, const float* g_j3, const float* g_or, const float* g_ot, const float* g_or2,
                  const float* g_ot2, int n_frames, int which, float* out, hipStream_t s);

// ---- network geometry (fixed by the reference confs; checked in hn_field_create) ---------
constexpr int H = 256;           // d_hidden == d_feature
constexpr int NT = H / 32;       // 8 row tiles of 32 neurons
constexpr int N_BONES = 21;
constexpr int PTS_FREQS = 10;    // v_multires
constexpr int OBJ_DIR_FREQS = 4; // r_multires, obj conf
constexpr int HAND_DIR_FREQS = 7;// r_multires, hand conf
constexpr int GRAD_FREQS = 4;    // grad_multires
constexpr int OBJ_IN = 63;
constexpr int BONE_FEAT = 66;
constexpr int HAND_IN = N_BONES * BONE_FEAT;  // 1386
constexpr int L3_OUT_OBJ = H - OBJ_IN;        // 193 -> 7 tiles

// K-step ("pair") spaces: one MFMA k-step consumes two input columns, one per
// lane half.  The order of the pairs is ours (baked into the packed weights).
constexpr int OBJ_X_STEPS = 32;   // 30 sin/cos pairs + (px,py) + (pz,0)
constexpr int VEC_STEPS = 16;     // enc4 of a 3-vector: 12 sin/cos pairs + (x,y) + (z,0) + 2 pad
constexpr int BONE_STEPS = 36;    // 33 pairs per bone + 3 pad
constexpr int HAND_X_STEPS = N_BONES * BONE_STEPS;   // 756
// reverse sweep over the hand features: groups of 4 bones = 144 pairs = 9 tiles
constexpr int BONE_GROUP = 4;
constexpr int GROUP_TILES = 9;
constexpr int N_GROUPS = 6;       // 5 full groups + 1 bone in the last

// accumulator-tile geometry of v_mfma_f32_32x32x2_f32: lane l holds column l&31;
// register r of lane half h = l>>5 holds row (r&3) + 8*(r>>2) + 4*h.
// MFMA shape of the f16x3 evaluation kernels (hn_mlp2.h, HN_MFMA16): which translation units / weight streams use
// v_mfma_f32_16x16x32_f16.  The adjoint kernels and the taped evaluation that feeds them stay on 32x32x16.
// Measured (round 2, profiles/r02/README.md): the hand evaluation kernel on 16x16x32 passes every parity test and takes
// 357.7 ms on the C2 frame against 278.5 ms on 32x32x16 -- half the filler budget per MFMA gap and 448 spilled
// registers -- so 32x32x16 stays the product; -DHN_HAND_EVAL_MFMA16=1 rebuilds the other arm.
#ifndef HN_HAND_EVAL_MFMA16
#define HN_HAND_EVAL_MFMA16 0
#endif
#ifndef HN_OBJ_EVAL_MFMA16
#define HN_OBJ_EVAL_MFMA16 0
#endif
__host__ __device__ inline int tile_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// One packed matrix: float4 fragments [out_tiles][steps/4][64 lanes].
struct PackedMat {
    const float4* w = nullptr;
    int out_tiles = 0;
    int steps = 0;   // k-steps (multiple of 4)
};

}  // namespace hn

// The opaque handle of the C ABI.
struct hn_field {
    int kind = 0;
    int precision = 0;
    float variance = 0.f;
    float inv_s = 0.f;
    const float* inv_s_dev = nullptr;   // hn_field_set_inv_s_device: inv_s lives on the device (borrowed; `inv_s` is then not meaningful)
    float scale = 1.f;
    void* blob = nullptr;       // one device allocation holding every packed array
    size_t blob_bytes = 0;
    // --- SDF network -------------------------------------------------------------------
    hn::PackedMat sdf_fwd[9];    // l = 0..7: W_l; [8]: rows 1..256 of W_8 (feature rows)
    hn::PackedMat sdf_skip;      // W_4[:, skip columns] over the input space (scaled 1/sqrt2)
    hn::PackedMat sdf_bwd[8];    // l = 1..7: W_l^T over hidden space ([0] unused)
    hn::PackedMat sdf_bwd_in0;   // W_0^T : rows = input space
    hn::PackedMat sdf_bwd_in4;   // W_4[:, skip]^T : rows = input space
    const float* sdf_bias[9] = {};   // tile-row order (padded to 32*tiles); [8] = feature rows
    const float* sdf_w8row = nullptr;// W_8[0, :] (256)
    float sdf_b8 = 0.f;              // b_8[0]
    // --- colour network ----------------------------------------------------------------
    hn::PackedMat col_in_x;      // lin0 columns over the point/feature input space
    hn::PackedMat col_in_d;      // lin0 columns over enc(view dir) (obj only)
    hn::PackedMat col_in_f;      // lin0 columns over the 256 feature vector
    hn::PackedMat col_in_g;      // lin0 columns over enc(gradient)
    hn::PackedMat col_fwd[4];    // l = 1..3 ([0] unused)
    const float* col_bias[4] = {};
    const float* col_wlast = nullptr;   // lin4: [3][256]
    float col_blast[3] = {0.f, 0.f, 0.f};
    // --- v2 (HN_PREC_F16X3): weight streams in consumption order (hn_pack2.hip) ---------
    void* v2_full = nullptr;     // sdf forward + reverse sweep + colour
    size_t v2_full_bytes = 0;
    void* v2_sdf = nullptr;      // sdf forward only (sampling passes)
    size_t v2_sdf_bytes = 0;
    void* v2_adj = nullptr;      // full evaluation followed by its adjoint (hn_field_eval_bwd)
    size_t v2_adj_bytes = 0;
    void* v2_tape = nullptr;     // the taped evaluation's program where it differs from v2_full (other MFMA shape); else NULL
    size_t v2_tape_bytes = 0;
    void* v2_adjonly = nullptr;  // the adjoint alone, from the tape a taped evaluation left (hn_render_dual / _bwd)
    size_t v2_adjonly_bytes = 0;
    // --- folded (weight-norm applied) weights and biases, row-major [out, in], for the adjoint (hn_field_bwd.hip)
    void* raw = nullptr;
    size_t raw_floats = 0;       // length of the retained block = length of a parameter-gradient vector (hn_field_param_floats)
    const float* raw_sdf_w[9] = {};
    const float* raw_sdf_b[9] = {};
    const float* raw_col_w[5] = {};
    const float* raw_col_b[5] = {};
    int sdf_out[9] = {}, sdf_in[9] = {}, col_out[5] = {}, col_in[5] = {};
    int sdf_ld[9] = {}, col_ld[5] = {};   // row pitch of the retained matrices (in_dim rounded up to a multiple of 4)
    int compact_far_field = 0;   // hn_field_set_compaction: the two-field renders evaluate only the samples with a live bone
    int single_pass = 0;         // HN_PREC_F16: the evaluation kernels run their hidden layers on one f16 MFMA per product
    int cull_far_field = 0;      // hn_field_set_culling: skip the chunks of bones whose mask is 0 for a whole workgroup
};
