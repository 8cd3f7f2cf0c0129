/* honerf.h -- C ABI of the MI355X-native HO-NeRF volume-rendering core.
 *
 * The reference (iscas3dv/HO-NeRF) is pure Python and has no FFI layer; its
 * boundary for this path is the Python object surface of utils/renderer.py,
 * utils/renderer_batch.py and utils/fields.py.  Every entry point below names
 * the reference function it replaces (paths relative to the reference
 * checkout); ho-nerf_amd/lib.py binds them with ctypes and
 * ho-nerf_amd/renderer*.py re-exposes the reference's classes on top.
 *
 * Conventions
 *   - every function returns HN_OK (0) or a negative HN_E* code; the message
 *     is available from hn_last_error() (thread-local);
 *   - all tensor arguments are DEVICE pointers to contiguous fp32 (indices:
 *     int64, to match torch) unless a parameter is documented as host;
 *   - the last argument is the hipStream_t to launch on (as void*); no entry
 *     point synchronises, allocates or frees device memory except
 *     hn_field_create / hn_field_destroy / hn_workspace_* (and hn_render_single_bwd
 *     on a hand field with hn_field_set_compaction, which reads one count back);
 *   - outputs and workspaces are caller-owned; sizes come from the
 *     *_workspace_bytes queries.
 */
#ifndef HONERF_H
#define HONERF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HN_VERSION 121 /* 0.1.3: bumped whenever a signature changes; lib.py refuses a library of another version */

#define HN_OK 0
#define HN_EINVAL (-1)   /* bad argument / unsupported shape */
#define HN_EHIP (-2)     /* a HIP runtime call failed */
#define HN_ENOMEM (-3)   /* workspace too small / allocation failed */

#define HN_FIELD_OBJ 0  /* SDFNetwork_OBJ + RenderingNetwork_OBJ (utils/fields.py:251-405) */
#define HN_FIELD_HAND 1 /* SDFNetwork + RenderingNetwork (utils/fields.py:56-240) */

#define HN_PREC_FP32 0   /* exact fp32: v_mfma_f32_32x32x2_f32 == fmaf chains */
#define HN_PREC_F16X3 1  /* fp16 hi/lo split operands, 3 x v_mfma_f32_32x32x16_f16 per product, fp32
                          * accumulate: fp32-equivalent results (22-bit operands) at 16/3 x the rate */
#define HN_PREC_F16 2    /* throughput mode: an HN_PREC_F16X3 field whose EVALUATION kernels (hn_field_sdf / hn_field_eval, the
                          * renders without a tape) run the hidden layers of both networks on ONE f16 MFMA per product (fp16
                          * operands, fp32 accumulation); the encodings, the feature layers, the last SDF layer, the alpha
                          * stage and every adjoint / taped kernel stay fp32-equivalent.  Error ~1e-3 (tests pin it):
                          * a secondary figure, never the parity path.  Hand fields; an obj field behaves as F16X3. */
/* OR-ed into `precision` at hn_field_create: pack the evaluation programs only, no adjoint weight streams.  For fields
 * that are re-packed every optimiser step (training, honerf_amd/training.py): their backward pass is
 * hn_field_param_bwd / hn_render_single_bwd, which work on the retained row-major matrices; hn_field_eval_bwd and
 * hn_render_dual_bwd still work on such a field (launch sequence instead of the fused adjoint kernels). */
#define HN_PACK_EVAL_ONLY 0x100

#define HN_MAX_LAYERS 9
#define HN_N_BONES 21

typedef void* hn_stream_t;         /* hipStream_t */
typedef struct hn_field hn_field;  /* opaque: packed, immutable weights of one field */

/* One MLP in the reference's state-dict layout (old-style weight-norm,
 * utils/fields.py:120-121): W[i,:] = weight_g[i] * weight_v[i,:] / ||weight_v[i,:]||.
 * weight_g[l] may be NULL: then weight_v[l] is the effective weight.
 * Pointers are DEVICE pointers. */
typedef struct {
    int n_layers;
    int out_dim[HN_MAX_LAYERS];
    int in_dim[HN_MAX_LAYERS];
    const float* weight_g[HN_MAX_LAYERS];
    const float* weight_v[HN_MAX_LAYERS];
    const float* bias[HN_MAX_LAYERS];
} hn_mlp_desc;

int hn_version(void);
const char* hn_last_error(void);

/* Number of compute units of the current device (0 if no device). */
int hn_device_cus(void);

/* ---- weights --------------------------------------------------------------------------
 * Folds weight-norm and re-lays every matrix in MFMA fragment order: once for the frozen
 * networks of rendering and pose fitting, after every optimiser step when training
 * (exp_runner.py:230-232).  `variance` is SingleVarianceNetwork.variance (utils/fields.py:243-249); `scale` is
 * SDFNetwork_OBJ.scale (:328).  The first pack of a field kind (and every HN_PREC_FP32 pack) lays the programs out on the
 * host and synchronises the stream; it also leaves a checked PLAN of the layout on the device, from which every later
 * f16x3 pack of that kind fills its programs with a memset and two launches each (~0.5 ms with HN_PACK_EVAL_ONLY) and
 * waits for nothing.
 * hn_field_destroy: the field's device blocks go to a size-keyed cache of the process and are handed to the next
 * hn_field_create of the same shape (a re-pack allocates nothing).  Work that uses the field may still be QUEUED on the stream
 * the next pack is issued on (a training loop destroys the old field while its backward pass is in flight): the new pack's
 * writes are ordered behind it.  Work in flight on ANOTHER stream must be waited for first.  Host staging is two pinned
 * buffers kept for the life of the process.  Packs of one process are serialised. */
int hn_field_create(int kind, const hn_mlp_desc* sdf, const hn_mlp_desc* color, float variance, float scale,
                    int precision, hn_field** out, hn_stream_t stream);
int hn_field_destroy(hn_field* f);
/* Returns the device blocks that destroyed fields left in the cache to the driver (hipFree: synchronises the device);
 * -> bytes released.  Optional: the cache is bounded by what this process's fields have held. */
size_t hn_release_cached_memory(void);
/* clip(exp(10 * variance), 1e-6, 1e6): utils/renderer.py:144. */
float hn_field_inv_s(const hn_field* f);
/* A TRAINED variance (exp_runner.py:107-110 steps `deviation_network.variance` with Adam every iteration) need not visit the host: with
 * `inv_s_dev` = a device float the caller keeps equal to clip(exp(10 variance), 1e-6, 1e6) (and alive while the field is used), the
 * single-field renders -- hn_render_single[_taped], hn_render_single_bwd[_taped] -- read inv_s from there at launch time and the
 * `variance` given to hn_field_create is ignored by them, so a training loop's per-iteration re-pack waits for nothing.  The two-field
 * renders refuse such a field (their fields are created once; HN_EINVAL).  NULL: back to the host value. */
int hn_field_set_inv_s_device(hn_field* f, const float* inv_s_dev);

/* Exact far-field early-out of the hand field (SURVEY B-11): a bone whose mask h = 1 - sigmoid(200 (v - cutoff))
 * (utils/fields.py:33-35) is exactly 0 in fp32 for every sample of a 128-sample workgroup contributes exactly 0
 * to lin0, the lin4 skip, colour lin0 and the gradient; with culling enabled the f16x3 kernels skip that bone's
 * weight chunks (results are bit-identical to the dense evaluation).  Off by default: throughput figures are
 * quoted dense.  Set it before launching work on the field; no effect on obj fields and on HN_PREC_FP32. */
int hn_field_set_culling(hn_field* f, int enabled);
/* Exact sample-level far-field skip of the hand field in the renders (hn_render_single, hn_render_dual / hn_render_dual_bwd; hand
 * fields of HN_PREC_F16X3, any number of frames, launches of >= 4096 samples): a sample whose 21 bone masks are all exactly
 * 0 has a constant sdf / colour, a zero gradient and contributes exactly 0 to every adjoint output, so the hand field is
 * evaluated on the compacted list of the other samples plus one far sample and the results are scattered back --
 * bit-identical outputs, 40 - 60 % fewer samples in a fitting step (one round of sample tiles on the chip instead of two),
 * ~5x on a 512 x 512 x 64 hand frame.  The stand-in ("far") sample is the first skipped sample of the launch itself, so it
 * is dead by the same predicate whatever the scene's scale.  Set it before sizing workspaces / tapes with the hn_render_*_workspace_bytes /
 * _tape_bytes queries (they grow by the compaction record) and do not change it between a render and its backward pass.
 * Off by default; the stand-alone evaluations (hn_field_sdf / hn_field_eval) and every throughput figure quoted as "dense"
 * never compact. */
int hn_field_set_compaction(hn_field* f, int enabled);
/* Measurement aid (bench.py): launches a kernel of nothing but v_mfma_f32_32x32x16_f16 on pseudo-random operands on every CU
 * (waves_per_simd = 1 or 2 workgroups of four waves per CU, `iters` x 32 MFMAs per wave) and returns the FLOP it issues in
 * *flop (may be NULL).  Timed by the caller on `stream`: the rate the power-limited matrix pipe sustains, which the roofline
 * entry of the bench line states beside the guide's nominal peak.  Not a product path. */
int hn_debug_mfma_probe(int waves_per_simd, int iters, double* flop, hn_stream_t stream);
/* Test hook for the XCD pacing of the f16x3 field kernels (image-sized launches: the 32 workgroups of an XCD meet at every
 * tile start, bounded spin): `members` > 0 registers that many members per XCD that never arrive, so that the first
 * meeting of every workgroup runs into its timeout and the launch continues unpaced -- results must be bit-identical.
 * 0 switches the hook off.  Process-wide; not for production use. */
int hn_debug_pace_phantom(int members);
/* Selection of the LATENCY-FORM kernels (hn_field2_hand_q.hip: the four waves of a workgroup share one 32-sample block and
 * split every layer's output tiles): by default sdf-only launches of a hand field of at most 2 x (CU count) blocks take it,
 * because they cannot fill the chip with whole 128-sample tiles.  Results are bit-identical either way.  max_blocks = 0:
 * never; > 0: up to that many blocks; -1: the default.  Process-wide; for tests and A/B timing. */
int hn_debug_quad_max_blocks(int max_blocks);
/* The importance rounds of hn_render_dual for small batches run cat_z_vals of the previous round, the gather of the hand's
 * compacted coarse sdf row, the column copy and the new sample positions inside the up_sample launch (one launch per round and
 * track).  on = 0: the separate launches (hn_merge, the scatter, hn_upsample, the copy, hn_sample_points) instead; results are
 * bit-identical either way.  Process-wide; for tests and A/B timing. */
int hn_debug_fused_rounds(int on);
/* Measurement aid (bench.py): while on, every EVALUATION launch of a field (hn_field_eval and the final evaluation inside
 * hn_render_single / hn_render_dual) is bracketed by two HIP events recorded on the stream it is launched on;
 * hn_debug_field_timer_read waits for the recorded pairs, returns their summed duration in milliseconds and their number, and
 * forgets them.  This is how the bench line times the dominant kernel INSIDE the timed steps (so that kernel time <= step time by
 * construction) instead of in a separate pass.  Process-wide; off by default; not a product path. */
int hn_debug_field_timer(int on);
int hn_debug_field_timer_read(double* total_ms, int* launches);
/* Samples the hand field's adjoint kernels DROPPED on the current device since the library was loaded (or since the last call with
 * reset != 0): a sample within ~2 mm of a bone's origin, whose adjoint quantities leave the fp16 fragments' range, gets g_pts = 0, no
 * share in the pose gradients and zero rows in the parameter-gradient signals, where the fp32 reference adds a huge finite value
 * (DESIGN.md 3.5).  Waits for the device: a diagnostic for tests, the bench line and a caller's logging, not a launch path. */
int hn_dropped_samples(unsigned long long* count, int reset);

/* ---- rays -----------------------------------------------------------------------------
 * _xy_to_ray_bundle (utils/utils.py:31-115): NDC xy -> unproject at depth 1 and
 * 2 -> d = normalize(p2-p1), o = p1 - d.  n_cams cameras, rays_per_cam rays each:
 * xy [n_cams*rays_per_cam, 2]; R [n_cams,3,3]; T [n_cams,3]; focal, principal
 * [n_cams,2] (PyTorch3D row-vector convention X_view = X_world R + T). */
int hn_ray_gen(const float* xy, const float* R, const float* T, const float* focal, const float* principal,
               int n_cams, int rays_per_cam, float* rays_o, float* rays_d, hn_stream_t stream);

/* convert_obj_to_local (utils/renderer.py:180-188, 424-432; renderer_batch.py:176-182):
 * o' = Ro (o - To), d' = Ro d.  n_frames poses, rays_per_frame rays each. */
int hn_obj_local_fwd(const float* rays_o, const float* rays_d, const float* Ro, const float* To, int n_frames,
                     int rays_per_frame, float* o_out, float* d_out, hn_stream_t stream);
/* Adjoint: given dL/do', dL/dd' -> dL/do, dL/dd [n,3] (may be NULL), dL/dRo [n_frames,3,3],
 * dL/dTo [n_frames,3] (both overwritten). */
int hn_obj_local_bwd(const float* rays_o, const float* rays_d, const float* Ro, const float* To, const float* g_o_out,
                     const float* g_d_out, int n_frames, int rays_per_frame, float* g_rays_o, float* g_rays_d,
                     float* g_Ro, float* g_To, hn_stream_t stream);

/* Coarse depths (utils/renderer.py:204-212): z[b,k] = near + (far-near) k/(n-1) +
 * (t_rand[b] - 0.5) (far-near)/n.  t_rand [B] in [0,1) comes from the caller's RNG.  near / far
 * are doubles (Python floats in the reference) and are rounded to fp32 where torch would. */
int hn_coarse_z(const float* t_rand, int n_rays, int n_samples, double near, double far, float* z,
                hn_stream_t stream);

/* Sample positions.  mid == 0: p = o + d z (utils/renderer.py:216).  mid == 1:
 * section mid-points p = o + d (z + dist/2), dist[k] = z[k+1]-z[k], last = sample_dist
 * (utils/renderer.py:119-123); `dists` [n_rays*n] is then written too (else may be NULL). */
int hn_sample_points(const float* rays_o, const float* rays_d, const float* z, int n_rays, int n, int mid,
                     float sample_dist, float* pts, float* dists, hn_stream_t stream);

/* Adjoint of hn_sample_points w.r.t. the rays (depths are sampled under no_grad): g_pts [n_rays*n,3] ->
 * g_rays_o, g_rays_d [n_rays,3] (overwritten). */
int hn_sample_points_bwd(const float* z, const float* g_pts, int n_rays, int n, int mid, float sample_dist,
                         float* g_rays_o, float* g_rays_d, hn_stream_t stream);

/* ---- hierarchical sampling ------------------------------------------------------------
 * NeuSRenderer.up_sample + sample_pdf(det=True) (utils/renderer.py:60-86, 10-37).
 * z, sdf [n_rays,k] -> z_new [n_rays,n_new]; inds (int64 [n_rays,n_new], may be
 * NULL) is the searchsorted(right=True) result.  Limits: 2 <= k <= 640, n_new >= 1 (the confs use k = 64..112,
 * n_new = 16; the forms tuned for them take k <= 256, n_new <= 64, a thread-per-ray form everything else); outside them
 * HN_EINVAL.  The renders accordingly take up to 640 depths per ray (two-field: per track, 1 024 in all). */
int hn_upsample(const float* z, const float* sdf, int n_rays, int k, int n_new, float inv_s, float* z_new,
                int64_t* inds, hn_stream_t stream);

/* cat_z_vals (utils/renderer.py:88-105): stable merge of sorted z [n_rays,k] with
 * sorted z_new [n_rays,m]; carries sdf / sdf_new when non-NULL; `index` (int64
 * [n_rays,k+m], may be NULL) is torch.sort's index into cat([z, z_new]).
 * sdf_row_stride_frames > 0 reproduces the batched renderer's quirk
 * (utils/renderer_batch.py:108-111, SURVEY B-1): with rays laid out
 * [frames, P], ray (f,p) gathers its SDF values from ray (0,p); pass P, or 0 for
 * the regular behaviour. */
int hn_merge(const float* z, const float* z_new, const float* sdf, const float* sdf_new, int n_rays, int k, int m,
             int quirk_rays_per_frame, float* z_out, float* sdf_out, int64_t* index, hn_stream_t stream);

/* Row-wise ascending sort of v [n_rays,n] (n <= 1024) -- the final torch.sort over
 * the concatenated depths of the two-field renderer (utils/renderer.py:498). */
int hn_sort_rows(const float* v, int n_rays, int n, float* out, hn_stream_t stream);

/* ---- fields ---------------------------------------------------------------------------
 * Hand fields take per-frame bone transforms: bt_inv [n_frames,21,4,4], T_pose
 * [n_frames,21,3]; point i belongs to frame i / pts_per_frame (one frame: pass
 * n_frames = 1, pts_per_frame = n_pts).  Obj fields ignore them (pass NULL). */
size_t hn_field_workspace_bytes(const hn_field* f, int n_pts);

/* .sdf() (utils/fields.py:158-160, 330-331): pts [n,3] -> sdf [n]; also the grid queries of extract_geometry
 * (utils/renderer.py:260-284, 537-564) in one launch. */
int hn_field_sdf(const hn_field* f, const float* pts, int n_pts, const float* bt_inv, const float* T_pose,
                 int n_frames, int pts_per_frame, float* sdf, void* workspace, size_t workspace_bytes,
                 hn_stream_t stream);

/* The three module calls of render_core / get_alpha_sample_color
 * (utils/renderer.py:130-142, 380-396): sdf network forward, its input
 * gradient (analytic, replaces autograd.grad of utils/fields.py:165-177, 336-347)
 * and the colour network.  pts [n,3]; dirs are per ray: rays_d [n/samples_per_ray,3].
 * -> sdf [n], grad [n,3], rgb [n,3]; feat (may be NULL) receives the 256-d
 * feature vector [n,256]. */
int hn_field_eval(const hn_field* f, const float* pts, const float* rays_d, int n_pts, int samples_per_ray,
                  const float* bt_inv, const float* T_pose, int n_frames, int pts_per_frame, float* sdf,
                  float* grad, float* rgb, float* feat, void* workspace, size_t workspace_bytes,
                  hn_stream_t stream);

/* ---- the modules one at a time (Public interface of L1, utils/fields.py) -------------------------------------
 * SDFNetwork.forward also returns xyz_feature, r, h (utils/fields.py:132-156; anerf_emb_point[_batch] :22-52):
 * pts [n,3] -> xyz_feature [n,1386], r [n,21,3] (may be NULL), h [n,21] (may be NULL). */
int hn_hand_features(const float* pts, int n_pts, const float* bt_inv, const float* T_pose, int n_frames,
                     int pts_per_frame, float* xyz_feature, float* r, float* h, hn_stream_t stream);

/* RenderingNetwork_OBJ.forward(points, view_dirs, feature_vectors, normals) (utils/fields.py:387-405) and
 * RenderingNetwork.forward(view_dirs, xyz_feature, feature_vectors, h, normals) (:222-240) with caller-supplied
 * inputs: x = points [n,3] (obj) or xyz_feature [n,1386] (hand; view_dirs is unused there and may be NULL),
 * view_dirs [n,3], feature_vectors [n,256], normals [n,3] -> rgb [n,3]. */
size_t hn_color_forward_workspace_bytes(const hn_field* f, int n_pts);
int hn_color_forward(const hn_field* f, const float* x, const float* view_dirs, const float* feature_vectors,
                     const float* normals, int n_pts, float* rgb, void* workspace, size_t workspace_bytes,
                     hn_stream_t stream);

/* Nearest candidate vertex, the cKDTree query of get_stable_loss_cross (utils/renderer_batch.py:355-358):
 * pts [n_verts,3]; query_mask, cand_mask [n_sets,n_verts] (bytes, non-zero = member).  For every query vertex of
 * set t the nearest candidate vertex of the same set is marked in selected [n_sets,n_verts] (zeroed here; the
 * reference's np.unique = a set); nearest (int32 [n_sets,n_verts], may be NULL) receives the index, -1 for
 * non-query vertices. */
int hn_nearest_masked(const float* pts, int n_verts, int n_sets, const unsigned char* query_mask,
                      const unsigned char* cand_mask, unsigned char* selected, int32_t* nearest, hn_stream_t stream);

/* ---- hand pose chain of the fitting loops ------------------------------------------------
 * fitting_single.py:206-226 (= fitting_video.py's per-frame chain): predicted joints, bone lengths and the refine
 * parameters -> refined joints -> `bone_transformation_inv`, i.e. convert_joints / transform_to_canonical /
 * PoseConverter.get_refine_3d_joint / rot6d_to_matrix / PoseConverter.forward (halo_util/converter_fit_batch.py:103-162,
 * 1109-1229; halo_util/utils.py:17-41; utils/utils.py:11-30) in one launch instead of ~4 000 torch operators.
 *   ori_pose [F,21,3]  the predicted joints, MANO order (ori_3d_pose);  bone_len [F,20]  cur_bone_length;
 *   is_right [F] bytes or NULL (all right hands, as the fitting scripts pass);
 *   params [F,36] = [joint_refine_angle 20 | palm_refine_angle 7 | palm_rot_refine 6 (row-major [3][2]) | palm_trans_refine 3]
 *   -> bt_inv [F,21,4,4] (bone_transformation_inv), joint_3d [F,21,3] (the refined joints, MANO order: the joint loss's
 *   input), and, when jac != NULL, jac [F,399,36] = d [bt_inv | joint_3d] / d params (forward-mode, exact).
 *   jac == NULL runs the values alone (a third of the time: what a fitting step's render waits for); bt_inv == joint_3d == NULL
 *   with jac != NULL the Jacobian alone (a caller that needs it later than the values asks for the two separately).
 * hn_pose_chain_bwd: g_params [F,36] = jac^T [g_bt_inv | g_joint_3d] (either gradient may be NULL = zero). */
#define HN_POSE_CHAIN_IN 36
#define HN_POSE_CHAIN_OUT 399
int hn_pose_chain(const float* ori_pose, const float* bone_len, const unsigned char* is_right, const float* params,
                  int n_frames, float* bt_inv, float* joint_3d, float* jac, hn_stream_t stream);
int hn_pose_chain_bwd(const float* jac, const float* g_bt_inv, const float* g_joint_3d, int n_frames, float* g_params,
                      hn_stream_t stream);

/* The rest of the fitting loops' pose side (fitting_single.py:213-217, 227-233, 119-122, 260), ~200 small torch operators
 * per step in the reference's form:
 * hn_rigid_pose: params [F,18] = [obj_rot_refine 6 (row-major [3][2]) | obj_trans_refine 3 | palm_rot_refine 6 | palm_trans_refine 3];
 *   bt_inv0 [F,21,4,4], joints0 [F,21,3] (used when with_palm != 0), Ro_pred [F,3,3], To_pred [F,3] ->
 *   out [F,412] = [bt_inv 336 = bt_inv0 G^-1 | joint_3d 63 = G joints0 | obj_r 9 = rot6d(obj_rot) Ro_pred | obj_t 3 |
 *   joint loss 1 = sum_j |joints0_j - joint_3d_j| / 21] with G p = R_palm (p - root) + root + T_palm, and, when jac != NULL,
 *   jac [F,412,18].  with_palm == 0: the object half only (entries 399 .. 410 of out / jac are written).
 * hn_verts_loss: pose_loss between the vertex sets of two rigid poses, loss[p] = mean_v |(Ra - Rb) v + (ta - tb)|, with
 *   its gradient w.r.t. (Ra, ta) (the gradient w.r.t. (Rb, tb) is its negative): Ra, Rb [P,3,3], ta, tb [P,3], verts [V,3].
 * hn_jacobian_vjp: out [F,n_in] = jac[F,n_out,n_in]^T g [F,n_out] (n_in <= 64): the backward pass of a dual-number op. */
#define HN_RIGID_POSE_IN 18
#define HN_RIGID_POSE_OUT 412
int hn_rigid_pose(const float* bt_inv0, const float* joints0, const float* Ro_pred, const float* To_pred,
                  const float* params, int n_frames, int with_palm, float* out, float* jac, hn_stream_t stream);
int hn_verts_loss(const float* Ra, const float* ta, const float* Rb, const float* tb, const float* verts, int n_verts,
                  int n_pairs, float* loss, float* gR, float* gt, hn_stream_t stream);
int hn_jacobian_vjp(const float* jac, const float* g, int n_frames, int n_out, int n_in, float* out, hn_stream_t stream);
/* The backward pass of the whole pose side of a fitting step (fitting_single.py:206-235 under loss.backward()) in one launch:
 * out [F,45] = [ [g_bt_inv 336 | g_joint_3d 63] . jac_h (hn_pose_chain's Jacobian [F,399,36]) | [g_obj_r 9 | g_obj_t 3] . rows
 * 399..410, columns 0..8 of jac_o (hn_rigid_pose's Jacobian [F,412,18], with_palm == 0) ]; a NULL upstream gradient is zero;
 * g_obj_r2 / g_obj_t2: second addends of the object's upstream gradients (the render's and the loss's shares, summed in the
 * kernel); `which`: 1 the hand columns 0..35 only, 2 the object columns 36..44 only, 3 both. */
/* The rows `rows` [n_rows] (int64) of a fitting_video sequence's six pose leaves (fitting_video.py:159-176; leaves6 = device pointers to
 * obj_rot [n,6], obj_trans [n,3], palm_rot [n,6], palm_trans [n,3], joint_refine_angle [n,20], palm_refine_angle [n,7], contiguous) as
 * the pose chain's input blocks, prm_hand [n_rows,36] (hn_pose_chain) and prm_obj [n_rows,18] (hn_rigid_pose; columns 9..17 zero), one
 * launch; and back: g [n_rows,45] = hn_pose_side_vjp's output scattered into those rows of six contiguous gradient blocks laid out one
 * behind the other in `out` (n_frames x 45 floats, zeroed by the caller): [n,6] [n,3] [n,6] [n,3] [n,20] [n,7].  Both take the number
 * of frames n the leaves hold: a row outside [0, n) is never dereferenced -- the gather writes NaN for it (the step's losses then say
 * so), the scatter drops it (torch's index_select / index_copy_ would have raised; a device-side index cannot without a sync). */
int hn_leaf_rows_gather(const float* const* leaves6, const long long* rows, int n_rows, int n_frames, float* prm_hand, float* prm_obj,
                        hn_stream_t stream);
int hn_leaf_rows_scatter(const float* g, const long long* rows, int n_rows, int n_frames, float* out, hn_stream_t stream);
int hn_pose_side_vjp(const float* jac_h, const float* jac_o, const float* g_bt_inv, const float* g_joint_3d, const float* g_obj_r,
                     const float* g_obj_t, const float* g_obj_r2, const float* g_obj_t2, int n_frames, int which, float* out,
                     hn_stream_t stream);

/* ---- SDF -> alpha, compositing --------------------------------------------------------
 * utils/renderer.py:147-161 (cos_anneal_ratio = 1): alpha [n] (clipped to [0,1]) and
 * c = sigmoid(prev_sdf * inv_s) [n] from sdf, grad, per-ray dirs and dists. */
int hn_alpha(const float* sdf, const float* grad, const float* rays_d, const float* dists, int n_pts,
             int samples_per_ray, float inv_s, float* alpha, float* c, hn_stream_t stream);

/* Single-field compositing (utils/renderer.py:163-169, 246-258): transmittance is
 * seeded with c[.,0] (not 1).  alpha, c [B,S]; rgb, grad [B,S,3] ->
 * color [B,3], weights [B,S] (may be NULL), weight_sum [B], weight_max [B],
 * eik_sum [1] += sum over samples of (||grad||-1)^2 (caller zeroes it, divides by B*S). */
int hn_composite1(const float* alpha, const float* c, const float* rgb, const float* grad, int n_rays, int S,
                  float* color, float* weights, float* weight_sum, float* weight_max, float* eik_sum,
                  hn_stream_t stream);

/* Two-field compositing (utils/renderer.py:512-524): T_k = prod_{j<k} (1-a_h+1e-7)(1-a_o+1e-7).
 * -> color [B,3], weight_sum [B], w_hand / w_obj [B,S] (may be NULL), eik_sum [2]
 * (hand, obj; accumulated, caller zeroes). */
int hn_composite2(const float* alpha_h, const float* rgb_h, const float* grad_h, const float* alpha_o,
                  const float* rgb_o, const float* grad_o, int n_rays, int S, float* color, float* weight_sum,
                  float* w_hand, float* w_obj, float* eik_sum, hn_stream_t stream);

/* Adjoint of hn_field_eval (what autograd runs through the two networks in the reference, including the second-order
 * path through `.gradient()`: utils/fields.py:165-177, 336-347 with create_graph=True; fitting_single.py:289-291).
 * g_sdf [n], g_grad [n,3], g_rgb [n,3] -> g_pts [n,3], g_rays_d [n/samples_per_ray,3] (may be NULL), and for hand
 * fields g_bt_inv [n_frames,21,4,4], g_T_pose [n_frames,21,3] (may be NULL).  The sweeps are specified in
 * oracle/field_bwd.py.  g_bt_inv / g_T_pose are ACCUMULATED into (zero them first).  Workspace: hn_field_bwd_workspace_bytes.
 * g_grad == g_rgb == NULL selects the adjoint of `.sdf()` alone (the hand SDF on object vertices of
 * get_stable_loss_cross, utils/renderer_batch.py:318-371): g_sdf -> g_pts and the pose gradients. */
size_t hn_field_bwd_workspace_bytes(const hn_field* f, int n_pts);
int hn_field_eval_bwd(const hn_field* f, const float* pts, const float* rays_d, int n_pts, int samples_per_ray,
                      const float* bt_inv, const float* T_pose, int n_frames, int pts_per_frame, const float* g_sdf,
                      const float* g_grad, const float* g_rgb, float* g_pts, float* g_rays_d, float* g_bt_inv,
                      float* g_T_pose, void* workspace, size_t workspace_bytes, hn_stream_t stream);

/* ---- adjoints of the two stages above (pose fitting back-propagates through them:
 * fitting_single.py:289-291, fitting_video.py:340-342).  Depths / dists carry no gradient (sampled under
 * no_grad, utils/renderer.py:215, 461).
 * hn_alpha_bwd: g_alpha, g_c [n] (g_c may be NULL) -> g_sdf [n], g_grad [n,3], g_rays_d [n/spr,3]
 * (zeroed here, then accumulated; may be NULL). */
int hn_alpha_bwd(const float* sdf, const float* grad, const float* rays_d, const float* dists, const float* g_alpha,
                 const float* g_c, int n_pts, int samples_per_ray, float inv_s, float* g_sdf, float* g_grad,
                 float* g_rays_d, hn_stream_t stream);
/* g_color [B,3], g_weight_sum [B] (may be NULL) -> g_alpha [B,S], g_c [B,S] (only column 0 is non-zero: the
 * seed), g_rgb [B,S,3]. */
int hn_composite1_bwd(const float* alpha, const float* c, const float* rgb, const float* g_color,
                      const float* g_weight_sum, int n_rays, int S, float* g_alpha, float* g_c, float* g_rgb,
                      hn_stream_t stream);
int hn_composite2_bwd(const float* alpha_h, const float* rgb_h, const float* alpha_o, const float* rgb_o,
                      const float* g_color, const float* g_weight_sum, int n_rays, int S, float* g_alpha_h,
                      float* g_rgb_h, float* g_alpha_o, float* g_rgb_o, hn_stream_t stream);

/* ---- whole renders ----------------------------------------------------------------------
 * NeuSRenderer.render (utils/renderer.py:190-258).  rays already in the field's frame
 * (obj: after hn_obj_local_fwd).  t_rand [B].  Outputs: color [B,3], cdf [B,S],
 * weight_sum [B], weight_max [B], gradient_error [1], z_vals [B,S] (may be NULL);
 * S = n_samples + n_importance.  s_val is 1/inv_s (host side). */
size_t hn_render_single_workspace_bytes(const hn_field* f, int n_rays, int n_samples, int n_importance);
int hn_render_single(const hn_field* f, const float* rays_o, const float* rays_d, const float* t_rand, int n_rays,
                     double near, double far, int n_samples, int n_importance, int up_sample_steps,
                     const float* bt_inv, const float* T_pose, float* color, float* cdf, float* weight_sum,
                     float* weight_max, float* gradient_error, float* z_vals, void* workspace,
                     size_t workspace_bytes, hn_stream_t stream);

/* (The hand and the object track of hn_render_dual are independent until their depths are merged and until the
 * compositing: they run side by side, the hand's on the given stream, the object's on a library-owned second stream of
 * the device, forked and joined with events -- the host is never blocked.)
 * NeuSRenderer_fitting.render (utils/renderer.py:434-535; frame-batched:
 * utils/renderer_batch.py:184-281).  World rays [n_frames*P,3]; Ro, To [n_frames,..];
 * bt_inv [n_frames,21,4,4]; T_pose [n_frames,21,3].  S = n_samples + 2 n_importance.
 * Outputs: color [N,3], weight_sum [N], sdf_hand, sdf_obj [N*S], grad_hand,
 * grad_obj [N*S,3], gradient_error [2] (hand, obj), z_vals [N,S] (may be NULL).
 * batch_quirk != 0 reproduces SURVEY B-1 (only meaningful for n_frames > 1). */
size_t hn_render_dual_workspace_bytes(const hn_field* hand, const hn_field* obj, int n_rays, int n_samples,
                                      int n_importance);
int hn_render_dual(const hn_field* hand, const hn_field* obj, const float* rays_o, const float* rays_d,
                   const float* t_rand, int n_frames, int rays_per_frame, double near, double far, int n_samples,
                   int n_importance, int up_sample_steps, const float* bt_inv, const float* T_pose, const float* Ro,
                   const float* To, int batch_quirk, float* color, float* weight_sum, float* sdf_hand,
                   float* sdf_obj, float* grad_hand, float* grad_obj, float* gradient_error, float* z_vals,
                   void* workspace, size_t workspace_bytes, void* tape, size_t tape_bytes, int flags, hn_stream_t stream);
/* flags of hn_render_dual / hn_render_dual_bwd (0: the reference's call as it stands):
 * HN_DUAL_RO_TRANSPOSED: `Ro` holds the matrix whose TRANSPOSE is the rotation of convert_obj_to_local -- the caller passes obj_r
 *   itself where the reference passes obj_r.T (fitting_single.py:250) -- and hn_render_dual_bwd's g_Ro is the gradient w.r.t. that
 *   matrix (same products in the same order; a transpose launch less on either side);
 * HN_DUAL_OBJ_POSE_ON_SIDE (hn_render_dual): Ro / To are produced on the device's second stream (hn_side_stream), where the
 *   object branch runs anyway: the object-local rays are made there instead of on `stream`;
 * HN_DUAL_BWD_NO_JOIN (hn_render_dual_bwd; needs g_rays_o = g_rays_d = NULL): `stream` does NOT wait for the object branch at the
 *   end: g_Ro / g_To are ready on the second stream, g_bt_inv / g_T_pose on `stream`.
 * Together they let a fitting loop keep the hand's and the object's halves of a step on two streams ACROSS steps (the hand's pose
 * chain and sampling track of step i + 1 start while the object's adjoint of step i is still running): honerf_amd.fitting. */
#define HN_DUAL_RO_TRANSPOSED 1
#define HN_DUAL_OBJ_POSE_ON_SIDE 2
#define HN_DUAL_BWD_NO_JOIN 4
/* The device's second stream (created on first use) and a device-side wait of one stream on another's current tail. */
int hn_side_stream(hn_stream_t* out);
int hn_stream_wait(hn_stream_t waiter, hn_stream_t on);
/* `tape` (may be NULL): when a backward pass will follow (the fitting loops), the final evaluation of both fields
 * keeps its tape -- activations, reverse-sweep values, feature fragments, per sample tile -- in this caller-owned buffer
 * of hn_render_dual_tape_bytes(hand, obj, n_rays, n_samples + 2 n_importance) bytes; hn_render_dual_bwd given the same
 * buffer then runs the adjoints alone instead of evaluating both fields a second time.  HN_PREC_F16X3 fields only
 * (0 bytes otherwise). */
size_t hn_render_dual_tape_bytes(const hn_field* hand, const hn_field* obj, int n_rays, int samples_per_ray);
/* With a tape, rgb_hand, rgb_obj [n_rays*S,3] and alpha_hand, alpha_obj [n_rays*S] of the final evaluation (what
 * hn_render_dual_bwd takes back) are written INTO the tape, contiguous and in that order, at this byte offset -- the
 * caller keeps the tape until the backward pass anyway, so nothing needs to be copied out of the workspace. */
size_t hn_render_dual_tape_aux_offset(const hn_field* hand, const hn_field* obj, int n_rays, int samples_per_ray);
/* After hn_render_dual the caller's workspace still holds what the final evaluation produced for compositing:
 * rgb_hand, rgb_obj [n_rays*S,3] and alpha_hand, alpha_obj [n_rays*S] (S = n_samples + 2 n_importance).  Byte offsets
 * of the four arrays into the workspace, in that order (same sizes and up_sample_steps as the render call) -- the backward pass of a fitting step re-uses them instead
 * of evaluating both fields again. */
int hn_render_dual_aux_offsets(const hn_field* hand, const hn_field* obj, int n_rays, int n_samples, int n_importance,
                               int up_sample_steps, size_t* offsets4);

/* Measurement aid (bench.py prices a fitting step on the samples it EXECUTES): with hn_field_set_compaction the hand field is
 * evaluated on the samples that have a live bone mask; the device-side count of each compacted launch (an int32: live
 * samples + 1, the stand-in) is kept at offsets2[0] bytes into the TAPE (the final evaluation of a taped render) and at
 * offsets2[1] bytes into the WORKSPACE (the coarse sdf pass of the importance sampling); (size_t)-1 where that launch of a
 * render of these sizes does not compact.  Valid after hn_render_dual until the buffers are re-used. */
int hn_render_dual_compact_offsets(const hn_field* hand, const hn_field* obj, int n_rays, int n_samples, int n_importance,
                                   int up_sample_steps, size_t* offsets2);

/* ---- the pose-only and loss side of a fitting_video window step in a handful of launches (hn_fit_window.hip) ----------------
 * hn_mat3_inverse / _bwd: torch.inverse(obj_r) of fitting_video.py:284 for n 3 x 3 matrices (adjugate), and its adjoint
 *   g_R = -Y^T g_out Y^T with Y the inverse.
 * get_stable_loss_cross (utils/renderer_batch.py:318-371) around the hand SDF:
 *   hn_stable_pts: every stride-th vertex of pts [n_frames, n_verts, 3] taken to the world with (obj_r, obj_t) (:319-321) ->
 *     pts_world [n_frames, ceil(n_verts / stride), 3]; p0 [ceil(n_verts / stride), 3] (may be NULL): frame 0's selected vertices
 *     in object coordinates (what the nearest-vertex query runs on, :352-353); hn_stable_pts_bwd: its adjoint w.r.t. obj_r, obj_t;
 *   hn_stable_value: sdf [n_frames, n_sel] of the hand on those points -> value[0] = the stable term, d_sdf [n_frames, n_sel] =
 *     d value / d sdf (inside sets, the nearest 'outside' vertex of every inside vertex -- the reference's cKDTree query --,
 *     weights: constants, as in the reference).  strict_reference != 0: the reference's 'outside' set (np.setdiff1d applied to the
 *     boolean mask).  n_frames <= 8, n_sel <= 1024; scratch: hn_stable_value_scratch_bytes bytes the caller
 *     ZEROES ONCE and keeps handing over (every launch leaves it ready for the next).
 * hn_field_tape_bytes / hn_field_eval_taped / hn_field_eval_bwd_taped: hn_field_eval that keeps its tape (HN_PREC_F16X3 fields
 *   with adjoint programs) and the adjoint alone from that tape (the taped pair hn_render_dual / _bwd use, for a caller's own
 *   points: the hand SDF on the object's vertices); g_grad / g_rgb: upstream gradients of all three outputs (zeros where unused).
 * hn_window_loss / _bwd: the whole loss of a window (fitting_video.py:285-334) in one launch each way:
 *   loss = w0 (colour + 0.5 mask) + w1 contact + w2 penetration + w3 joint + w4 verts + w5 smooth + w6 stable  (weights7: host
 *   array; the reference: {0.5, 30, 20, 30, 20, 50, 100}), joint / verts / smooth over the window's n_frames (<= 8) frames as
 *   pose_loss (mean) of fitting_video.py:123-126, :310-321; anchor bit 0 / bit 1: the window starts / ends the sequence and the
 *   smoothness term is anchored to the prediction there (:312-320); stable: device scalar or NULL.
 *   terms10 = {loss, colour, mask, contact, penetration, joint, verts, w5 smooth, w6 stable, 0}; g_joint [n_frames,21,3], gR
 *   [n_frames,9], gt [n_frames,3]: d (weighted pose part) / d (joint_3d, obj_r, obj_t); scratch as hn_fit_step_loss's. */
int hn_mat3_inverse(const float* R, int n, float* out, hn_stream_t stream);
int hn_mat3_inverse_bwd(const float* R_inv, const float* g_out, int n, float* g_R, hn_stream_t stream);
int hn_stable_pts(const float* pts, int n_frames, int n_verts, int stride, const float* obj_r, const float* obj_t, float* pts_world, float* p0,
                  hn_stream_t stream);
int hn_stable_pts_bwd(const float* pts, int n_frames, int n_verts, int stride, const float* g_pts_world, float* g_obj_r, float* g_obj_t,
                      hn_stream_t stream);
size_t hn_stable_value_scratch_bytes(int n_frames, int n_sel);
int hn_stable_value(const float* sdf, const float* p0, int n_frames, int n_sel, int strict_reference, float* value, float* d_sdf, void* scratch,
                    size_t scratch_bytes, hn_stream_t stream);
size_t hn_field_tape_bytes(const hn_field* f, int n_pts);
int hn_field_eval_taped(const hn_field* f, const float* pts, const float* rays_d, int n_pts, int samples_per_ray, const float* bt_inv,
                        const float* T_pose, int n_frames, int pts_per_frame, float* sdf, float* grad, float* rgb, void* workspace,
                        size_t workspace_bytes, void* tape, size_t tape_bytes, hn_stream_t stream);
int hn_field_eval_bwd_taped(const hn_field* f, const float* pts, const float* rays_d, int n_pts, int samples_per_ray, const float* bt_inv,
                            const float* T_pose, int n_frames, int pts_per_frame, const float* g_sdf, const float* g_grad, const float* g_rgb,
                            const float* grad, const float* rgb, const void* tape, float* g_pts, float* g_rays_d, float* g_bt_inv, float* g_T_pose,
                            void* workspace, size_t workspace_bytes, hn_stream_t stream);
size_t hn_window_loss_scratch_bytes(int n_rays, int n_samples);
int hn_window_loss(const float* color, const float* weight_sum, const float* true_rgb, const float* true_mask, int n_rays, const float* sdf_hand,
                   const float* sdf_obj, int n_samples, const float* joint_3d, const float* joint3d_pred, int n_frames, const float* obj_r,
                   const float* obj_t, const float* Ro_pred, const float* To_pred, const float* verts, int n_verts, const float* stable, int anchor,
                   const float* weights7, void* scratch, size_t scratch_bytes, float* sums6, float* terms10, float* g_joint, float* gR, float* gt,
                   hn_stream_t stream);
int hn_window_loss_bwd(const float* color, const float* weight_sum, const float* true_rgb, const float* true_mask, int n_rays, const float* sdf_hand,
                       const float* sdf_obj, int n_samples, const float* sums6, const float* g_loss, const float* weights7, const float* g_joint,
                       const float* gR, const float* gt, int n_frames, float* g_color, float* g_weight_sum, float* g_sdf_hand, float* g_sdf_obj,
                       float* g_joint_out, float* gR_out, float* gt_out, float* g_stable, hn_stream_t stream);

/* The render-dependent loss terms of one fitting step (fitting_single.py:251-283; fitting_video.py:285-309): the sums
 * behind colour L1, mask BCE, contact and penetration in one launch, and their gradients w.r.t. the render outputs in
 * another.  color [R,3], weight_sum [R], true_rgb [R,3], true_mask [R]; sdf_hand, sdf_obj [n_samples] (both NULL: no
 * interaction terms).  sums6 (zeroed here) = {sum |(color - true_rgb) mask| / R, sum BCE(clip(weight_sum, 1e-3,
 * 1 - 1e-3), mask) / R, contact sum, contact count, penetration sum, penetration count}; the losses are sums6[0],
 * sums6[1], sums6[2] / (sums6[3] + 1e-9), sums6[4] / (sums6[5] + 1e-9).  g4: device scalars, upstream gradients of those four
 * losses. */
int hn_fit_loss_sums(const float* color, const float* weight_sum, const float* true_rgb, const float* true_mask, int n_rays,
                     const float* sdf_hand, const float* sdf_obj, int n_samples, float* sums6, hn_stream_t stream);
int hn_fit_loss_grads(const float* color, const float* weight_sum, const float* true_rgb, const float* true_mask, int n_rays,
                      const float* sdf_hand, const float* sdf_obj, int n_samples, const float* sums6, const float* g4,
                      float* g_color, float* g_weight_sum, float* g_sdf_hand, float* g_sdf_obj, hn_stream_t stream);

/* The whole loss of a fitting_single step from those sums (fitting_single.py:251-288), one launch:
 *   loss = w0 (colour + 0.5 mask) + w1 contact + w2 penetration + w3 joint + w4 verts,
 * joint = sum_j |joint3d_pred_j - joint_3d_j| / n_joints (pose_loss, :119-122), verts = verts_loss[0] (hn_verts_loss).
 * weights5 is a HOST array {w0..w4} ({1, 30, 20, 30, 20} for fit type 12; {1, 0, 0, 100, 5} for fit type 1).
 * terms8 (device) = {loss, colour, mask, contact, penetration, joint, verts, 0}; g_joint [n_joints,3] = d joint / d joint_3d.
 * hn_fit_total_bwd: the upstream gradient of the loss (device scalar g_loss) -> g4 for hn_fit_loss_grads and the scaled
 * pose-side gradients g_joint_out = g w3 g_joint, gR_out [9] = g w4 gR, gt_out [3] = g w4 gt (gR, gt from hn_verts_loss). */
int hn_fit_total(const float* sums6, const float* verts_loss, const float* joint_3d, const float* joint3d_pred, int n_joints,
                 const float* weights5, float* terms8, float* g_joint, hn_stream_t stream);
int hn_fit_total_bwd(const float* g_loss, const float* weights5, const float* g_joint, const float* gR, const float* gt, int n_joints,
                     float* g4, float* g_joint_out, float* gR_out, float* gt_out, hn_stream_t stream);
/* SingleVarianceNetwork on device scalars (utils/fields.py:248-249, utils/renderer.py:144): inv_s = clip(exp(10 variance), 1e-6, 1e6), and the
 * chain rule of a backward pass, g_variance = g_inv_s x (10 inv_s inside the clip range, else 0) -- for hn_field_set_inv_s_device fields, whose
 * variance never visits the host. */
int hn_variance_to_inv_s(const float* variance, float* inv_s, hn_stream_t stream);
int hn_variance_chain(const float* g_inv_s, const float* inv_s, float* g_variance, hn_stream_t stream);
/* The loss of a training iteration (exp_runner.py:202-212 without the VGG term, which stays a torch module on color_fine) as ONE launch
 * forward and ONE backward: m = (true_mask > 0.5), mask_sum = sum m + 1e-5, colour = sum |(color - true_rgb) m| / mask_sum, mask =
 * binary_cross_entropy(clip(weight_sum, 1e-3, 1 - 1e-3), m), loss = colour + mask_weight mask + igr_weight gradient_error.
 * terms6 (device) = loss, colour, mask, gradient_error, psnr (:206), mask_sum; sums in a fixed order.  hn_train_loss_bwd: the upstream
 * gradient of the loss (device scalar) -> g_color [n_rays,3], g_weight_sum [n_rays], g_gradient_error [1]. */
int hn_train_loss(const float* color, const float* weight_sum, const float* gradient_error, const float* true_rgb, const float* true_mask, int n_rays,
                  float igr_weight, float mask_weight, float* terms6, hn_stream_t stream);
int hn_train_loss_bwd(const float* color, const float* weight_sum, const float* true_rgb, const float* true_mask, int n_rays, const float* terms6,
                      const float* g_loss, float igr_weight, float mask_weight, float* g_color, float* g_weight_sum, float* g_gradient_error,
                      hn_stream_t stream);
/* The same loss (fitting_single.py:251-288) as ONE launch forward and ONE backward -- what the fitting loop runs: the sums, the
 * vertex loss of the pose pair (Ra, ta) / (Rb, tb) over `verts` [n_verts,3] (fitting_single.py:232-233), the joint loss and the
 * weighted total in hn_fit_step_loss (sums6 [6], terms8 [8], g_joint [n_joints,3], gR [9], gt [3] as above; the sums are
 * reduced in a fixed order, so two runs give the same terms bit for bit); hn_fit_step_loss_bwd takes the upstream gradient of the
 * loss (device scalar) and returns the gradients w.r.t. the render outputs and the scaled pose-side gradients.  `scratch`:
 * hn_fit_step_loss_scratch_bytes(n_rays, n_samples) bytes the caller ZEROES ONCE and then keeps handing over (every launch
 * leaves it ready for the next); weights5: host array. */
size_t hn_fit_step_loss_scratch_bytes(int n_rays, int n_samples);
int hn_fit_step_loss(const float* color, const float* weight_sum, const float* true_rgb, const float* true_mask, int n_rays, const float* sdf_hand,
                     const float* sdf_obj, int n_samples, const float* joint_3d, const float* joint3d_pred, int n_joints, const float* Ra, const float* ta,
                     const float* Rb, const float* tb, const float* verts, int n_verts, const float* weights5, void* scratch, size_t scratch_bytes,
                     float* sums6, float* terms8, float* g_joint, float* gR, float* gt, hn_stream_t stream);
int hn_fit_step_loss_bwd(const float* color, const float* weight_sum, const float* true_rgb, const float* true_mask, int n_rays, const float* sdf_hand,
                         const float* sdf_obj, int n_samples, const float* sums6, const float* g_loss, const float* weights5, const float* g_joint,
                         const float* gR, const float* gt, int n_joints, float* g_color, float* g_weight_sum, float* g_sdf_hand, float* g_sdf_obj,
                         float* g_joint_out, float* gR_out, float* gt_out, hn_stream_t stream);
/* The same two launches for SEVERAL independent frames at once (fitting_single frames a rank fits side by side: every frame is its own
 * loss with its own normalisation, fitting_single.py:251-288): every per-frame array holds n_frames (<= 16) planes, frame after frame
 * (color [n_frames x n_rays,3], sdf [n_frames x n_samples], joint_3d [n_frames,n_joints,3], Ra [n_frames,9], sums6 [n_frames,6], terms8
 * [n_frames,8], ...), `verts` / `n_verts` are HOST arrays of the frames' vertex pointers and counts, `scratch` holds n_frames x
 * hn_fit_step_loss_scratch_bytes(n_rays, n_samples) bytes (zeroed once), g_loss is the one upstream scalar.  Frame f's results are
 * bit for bit those of hn_fit_step_loss / _bwd on that frame's planes: its own blocks, partial-sum slots and summation order. */
int hn_fit_step_loss_frames(int n_frames, const float* color, const float* weight_sum, const float* true_rgb, const float* true_mask, int n_rays,
                            const float* sdf_hand, const float* sdf_obj, int n_samples, const float* joint_3d, const float* joint3d_pred, int n_joints,
                            const float* Ra, const float* ta, const float* Rb, const float* tb, const float* const* verts, const int* n_verts,
                            const float* weights5, void* scratch, size_t scratch_bytes, float* sums6, float* terms8, float* g_joint, float* gR, float* gt,
                            hn_stream_t stream);
int hn_fit_step_loss_bwd_frames(int n_frames, const float* color, const float* weight_sum, const float* true_rgb, const float* true_mask, int n_rays,
                                const float* sdf_hand, const float* sdf_obj, int n_samples, const float* sums6, const float* g_loss,
                                const float* weights5, const float* g_joint, const float* gR, const float* gt, int n_joints, float* g_color,
                                float* g_weight_sum, float* g_sdf_hand, float* g_sdf_obj, float* g_joint_out, float* gR_out, float* gt_out,
                                hn_stream_t stream);

/* torch.optim.Adam's step (defaults: betas as given, no weight decay, no amsgrad) over up to 16 small parameter blocks
 * with one learning rate each, ONE launch: the six pose-parameter groups of fitting_single.py:191-199 /
 * fitting_video.py:177-185.  All pointer arrays are HOST arrays of device pointers (params, grads, the two moment
 * buffers -- caller-owned, zero-initialised), sizes in floats; steps[i] = 1, 2, ...: the number of updates block i has
 * received including this one (torch keeps the count per parameter: a block without a gradient is skipped). */
int hn_adam_step(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                 const int* sizes, const float* lr, float beta1, float beta2, float eps, const int* steps, hn_stream_t stream);

/* Backward pass of hn_render_dual: what loss.backward() runs through NeuSRenderer_fitting.render in the fitting loops
 * (fitting_single.py:289-291, fitting_video.py:340-342; autograd through utils/renderer.py:434-535).  Depths carry no
 * gradient (utils/renderer.py:461: sampled under no_grad).  Inputs: the render's inputs, its final depths z_vals
 * [N,S] and per-sample results (sdf_*, grad_* as returned; rgb_*, alpha_* from hn_render_dual_aux_offsets), and the
 * upstream gradients of every output: g_color [N,3], g_weight_sum [N] (may be NULL), g_sdf_* [N*S], g_grad_* [N*S,3],
 * g_gradient_error [2] (hand, obj) -- each may be NULL.  Outputs (overwritten): g_rays_o, g_rays_d [N,3] (both NULL: the
 * world rays' gradients are not wanted -- the fitting loops' rays come from fixed cameras -- and the launches that only
 * serve them are skipped), g_bt_inv [n_frames,21,4,4], g_T_pose [n_frames,21,3], g_Ro [n_frames,3,3], g_To [n_frames,3]
 * (the last two accumulated with float atomics: equal to rounding between runs).  The hand and the object branch
 * run side by side on the given stream and on a library-owned second stream of the device (event fork / join; the host
 * is never blocked).  `tape`: the buffer the forward hn_render_dual call filled (same rays, depths and fields), or NULL
 * (then both fields are evaluated again inside their adjoint kernels). */
size_t hn_render_dual_bwd_workspace_bytes(const hn_field* hand, const hn_field* obj, int n_rays, int samples_per_ray);
int hn_render_dual_bwd(const hn_field* hand, const hn_field* obj, const float* rays_o, const float* rays_d, int n_frames,
                       int rays_per_frame, int samples_per_ray, float sample_dist, const float* bt_inv, const float* T_pose,
                       const float* Ro, const float* To, const float* z_vals, const float* sdf_hand, const float* grad_hand,
                       const float* rgb_hand, const float* alpha_hand, const float* sdf_obj, const float* grad_obj,
                       const float* rgb_obj, const float* alpha_obj, const float* g_color, const float* g_weight_sum,
                       const float* g_sdf_hand, const float* g_sdf_obj, const float* g_grad_hand, const float* g_grad_obj,
                       const float* g_gradient_error, float* g_rays_o, float* g_rays_d, float* g_bt_inv, float* g_T_pose,
                       float* g_Ro, float* g_To, void* workspace, size_t workspace_bytes, const void* tape, int flags,
                       hn_stream_t stream);

/* ---- parameter gradients: training the networks (exp_runner.py:208-242 `loss.backward()` into
 * sdf_network / color_network / deviation_network, optimiser step at :230-232).
 * A field's trainable state as this library sees it is the FOLDED weights W_l = g_l v_l / |v_l| (old weight-norm API,
 * utils/fields.py:113-121, 296-306) and biases, row-major [out, ld] per layer (ld = in rounded up to 4), in one flat
 * block of hn_field_param_floats(f) floats; hn_field_param_offset gives a layer's place in it (net 0 = SDF network,
 * layers 0..8; net 1 = colour network, layers 0..4).  A gradient vector has the same layout.  The weight-norm chain rule
 * (d/d weight_g, d/d weight_v) and the optimiser are the caller's: element-wise over these blocks
 * (honerf_amd/training.py does it with torch's Adam, as exp_runner.py:97-104).
 *
 * hn_field_param_bwd: the adjoint of hn_field_eval (same arguments as hn_field_eval_bwd) that also ACCUMULATES
 * d loss / d (folded weights, biases) into g_params (zero it first); g_pts etc. as hn_field_eval_bwd.  Both the
 * first-order path and the path through `.gradient()` (create_graph=True in the reference) are included.
 *
 * hn_render_single_bwd: the whole backward pass of NeuSRenderer.render (utils/renderer.py:190-258) at the depths
 * z_vals [B,S] the forward pass returned (sampling is under no_grad, :215): g_color [B,3], g_weight_sum [B] (may be
 * NULL), g_gradient_error [1] (may be NULL) -> g_params (accumulated), g_inv_s [1] (d/d inv_s; inv_s = exp(10 variance),
 * utils/fields.py SingleVarianceNetwork; may be NULL), g_rays_o / g_rays_d [B,3] (in the field's frame; may be NULL),
 * g_bt_inv [21,4,4] / g_T_pose [21,3] (hand; may be NULL).  rays are in the field's frame (obj: after
 * hn_obj_local_fwd).  On a hand field with hn_field_set_compaction the launch sequence runs on the samples with a live bone
 * mask plus one far sample that carries the summed upstream gradients of all the others (exact: they share its input); the
 * sequence is sized on the host, so this call then waits for `stream` once (the live count is read back). */
size_t hn_field_param_floats(const hn_field* f);
int hn_field_param_offset(const hn_field* f, int net, int layer, size_t* w_off, size_t* b_off, int* out_dim, int* in_dim,
                          int* ld);
int hn_field_param_bwd(const hn_field* f, const float* pts, const float* rays_d, int n_pts, int samples_per_ray,
                       const float* bt_inv, const float* T_pose, int n_frames, int pts_per_frame, const float* g_sdf,
                       const float* g_grad, const float* g_rgb, float* g_params, float* g_pts, float* g_rays_d,
                       float* g_bt_inv, float* g_T_pose, void* workspace, size_t workspace_bytes, hn_stream_t stream);
/* The weight-norm chain rule for all 14 layers (utils/fields.py:113-121; torch's `_weight_norm` backward): from g_params
 * (the layout above) and the current parameters `sdf` / `color` (as given to hn_field_create) to the gradients of
 * weight_g [out,1], weight_v [out,in] and bias [out], written through the pointers of `g_sdf` / `g_color` (same struct;
 * its pointers are OUTPUTS here).  A layer without weight_g (plain nn.Linear) gets d/d weight = dW in weight_v. */
int hn_weight_norm_bwd(const hn_field* f, const hn_mlp_desc* sdf, const hn_mlp_desc* color, const float* g_params,
                       const hn_mlp_desc* g_sdf, const hn_mlp_desc* g_color, hn_stream_t stream);
size_t hn_render_single_bwd_workspace_bytes(const hn_field* f, int n_rays, int samples_per_ray);
int hn_render_single_bwd(const hn_field* f, const float* rays_o, const float* rays_d, int n_rays, int samples_per_ray,
                         float sample_dist, const float* bt_inv, const float* T_pose, const float* z_vals,
                         const float* g_color, const float* g_weight_sum, const float* g_gradient_error, float* g_params,
                         float* g_inv_s, float* g_rays_o, float* g_rays_d, float* g_bt_inv, float* g_T_pose,
                         void* workspace, size_t workspace_bytes, hn_stream_t stream);
/* The same pair with the evaluation's tape kept between the passes (a training iteration: exp_runner.py:196-232 renders, then calls
 * loss.backward()).  hn_render_single_taped = hn_render_single whose final evaluation of the n_rays x (n_samples + n_importance) samples
 * keeps its tape and outputs in the caller's block `tape` (hn_render_single_tape_bytes bytes; 0: this field's backward pass takes
 * no tape -- not HN_PREC_F16X3, or packed without its tape programs -- use the plain pair); hn_render_single_bwd_taped =
 * hn_render_single_bwd that reads that block instead of evaluating the field again.  The gradients are those of the plain pair (whose
 * backward pass runs the same taped evaluation itself; float atomics aside); the render outputs equal hn_render_single's to the bit for
 * an object field and to fp32 rounding for a hand field (its taped kernel contracts d sdf / d pts through the per-bone sums the
 * adjoint needs again: the same sums in another order).  The block must be the one the forward call of the SAME rays, poses,
 * weights and compaction setting filled. */
size_t hn_render_single_tape_bytes(const hn_field* f, int n_rays, int samples_per_ray);
int hn_render_single_taped(const hn_field* f, const float* rays_o, const float* rays_d, const float* t_rand, int n_rays, double near,
                           double far, int n_samples, int n_importance, int up_sample_steps, const float* bt_inv, const float* T_pose,
                           float* color, float* cdf, float* weight_sum, float* weight_max, float* gradient_error, float* z_vals,
                           void* tape, size_t tape_bytes, void* workspace, size_t workspace_bytes, hn_stream_t stream);
int hn_render_single_bwd_taped(const hn_field* f, const float* rays_o, const float* rays_d, int n_rays, int samples_per_ray,
                               float sample_dist, const float* bt_inv, const float* T_pose, const float* z_vals, const float* g_color,
                               const float* g_weight_sum, const float* g_gradient_error, float* g_params, float* g_inv_s,
                               float* g_rays_o, float* g_rays_d, float* g_bt_inv, float* g_T_pose, const void* tape, size_t tape_bytes,
                               void* workspace, size_t workspace_bytes, hn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HONERF_H */
