"""ctypes binding of libhonerf.so (include/honerf.h).

The library is built in-tree by ``make -C ho-nerf_amd/csrc`` (or
``__graft_entry__.build()``).  There is NO fallback: if the shared object is
missing or a call fails, a RuntimeError is raised -- the product path never
routes through a CPU or eager-PyTorch substitute.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('HONERF_LIB') or os.path.join(_HERE, 'libhonerf.so')   # HONERF_LIB: an A/B build (tools/)

HN_FIELD_OBJ = 0
HN_FIELD_HAND = 1
HN_PREC_FP32 = 0
HN_PREC_F16X3 = 1
HN_PREC_F16 = 2           # single-pass throughput mode of the evaluation kernels (include/honerf.h)
HN_DUAL_RO_TRANSPOSED, HN_DUAL_OBJ_POSE_ON_SIDE, HN_DUAL_BWD_NO_JOIN = 1, 2, 4      # flags of hn_render_dual / _bwd
HN_PACK_EVAL_ONLY = 0x100     # no adjoint weight streams (fields re-packed every training step)
PRECISIONS = {'fp32': HN_PREC_FP32, 'f16x3': HN_PREC_F16X3, 'f16': HN_PREC_F16}
# 'f16x3': fp16 hi/lo split operands on the f16 MFMA, fp32-equivalent results (the default);
# 'fp32': the exact-fp32 MFMA path (v_mfma_f32_32x32x2_f32), 5x slower, kept as a second opinion
DEFAULT_PRECISION = os.environ.get('HONERF_PRECISION', 'f16x3')
HN_MAX_LAYERS = 9
HN_VERSION = 121          # the include/honerf.h revision SIGNATURES below was written for

c_f = ctypes.c_void_p     # device float*
c_i = ctypes.c_int
c_sz = ctypes.c_size_t
c_fl = ctypes.c_float
c_db = ctypes.c_double
c_vp = ctypes.c_void_p


class MlpDesc(ctypes.Structure):
    _fields_ = [
        ('n_layers', c_i),
        ('out_dim', c_i * HN_MAX_LAYERS),
        ('in_dim', c_i * HN_MAX_LAYERS),
        ('weight_g', c_vp * HN_MAX_LAYERS),
        ('weight_v', c_vp * HN_MAX_LAYERS),
        ('bias', c_vp * HN_MAX_LAYERS),
    ]


# name -> (restype, argtypes); mirrors include/honerf.h one to one
SIGNATURES = {
    'hn_version': (c_i, []),
    'hn_last_error': (ctypes.c_char_p, []),
    'hn_device_cus': (c_i, []),
    'hn_field_create': (c_i, [c_i, ctypes.POINTER(MlpDesc), ctypes.POINTER(MlpDesc), c_fl, c_fl, c_i,
                              ctypes.POINTER(c_vp), c_vp]),
    'hn_field_destroy': (c_i, [c_vp]),
    'hn_field_inv_s': (c_fl, [c_vp]),
    'hn_field_set_culling': (c_i, [c_vp, c_i]),
    'hn_field_set_compaction': (c_i, [c_vp, c_i]),
    'hn_debug_pace_phantom': (c_i, [c_i]),
    'hn_debug_mfma_probe': (c_i, [c_i, c_i, ctypes.POINTER(ctypes.c_double), c_vp]),
    'hn_debug_quad_max_blocks': (c_i, [c_i]),
    'hn_debug_fused_rounds': (c_i, [c_i]),
    'hn_debug_field_timer': (c_i, [c_i]),
    'hn_debug_field_timer_read': (c_i, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_i)]),
    'hn_dropped_samples': (c_i, [ctypes.POINTER(ctypes.c_ulonglong), c_i]),
    'hn_field_set_inv_s_device': (c_i, [c_vp, c_f]),
    'hn_ray_gen': (c_i, [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_f, c_f, c_vp]),
    'hn_obj_local_fwd': (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_f, c_f, c_vp]),
    'hn_obj_local_bwd': (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_f, c_f, c_f, c_f, c_vp]),
    'hn_coarse_z': (c_i, [c_f, c_i, c_i, c_db, c_db, c_f, c_vp]),
    'hn_sample_points': (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_fl, c_f, c_f, c_vp]),
    'hn_sample_points_bwd': (c_i, [c_f, c_f, c_i, c_i, c_i, c_fl, c_f, c_f, c_vp]),
    'hn_upsample': (c_i, [c_f, c_f, c_i, c_i, c_i, c_fl, c_f, c_vp, c_vp]),
    'hn_merge': (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_f, c_f, c_vp, c_vp]),
    'hn_sort_rows': (c_i, [c_f, c_i, c_i, c_f, c_vp]),
    'hn_field_workspace_bytes': (c_sz, [c_vp, c_i]),
    'hn_field_sdf': (c_i, [c_vp, c_f, c_i, c_f, c_f, c_i, c_i, c_f, c_vp, c_sz, c_vp]),
    'hn_field_eval': (c_i, [c_vp, c_f, c_f, c_i, c_i, c_f, c_f, c_i, c_i, c_f, c_f, c_f, c_f, c_vp, c_sz, c_vp]),
    'hn_field_bwd_workspace_bytes': (c_sz, [c_vp, c_i]),
    'hn_field_eval_bwd': (c_i, [c_vp, c_f, c_f, c_i, c_i, c_f, c_f, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_vp, c_sz, c_vp]),
    'hn_hand_features': (c_i, [c_f, c_i, c_f, c_f, c_i, c_i, c_f, c_f, c_f, c_vp]),
    'hn_color_forward_workspace_bytes': (c_sz, [c_vp, c_i]),
    'hn_color_forward': (c_i, [c_vp, c_f, c_f, c_f, c_f, c_i, c_f, c_vp, c_sz, c_vp]),
    'hn_nearest_masked': (c_i, [c_f, c_i, c_i, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'hn_pose_chain': (c_i, [c_f, c_f, c_vp, c_f, c_i, c_f, c_f, c_f, c_vp]),
    'hn_pose_chain_bwd': (c_i, [c_f, c_f, c_f, c_i, c_f, c_vp]),
    'hn_rigid_pose': (c_i, [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_f, c_f, c_vp]),
    'hn_verts_loss': (c_i, [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_f, c_f, c_f, c_vp]),
    'hn_leaf_rows_gather': (c_i, [ctypes.POINTER(c_vp), c_vp, c_i, c_i, c_f, c_f, c_vp]),
    'hn_leaf_rows_scatter': (c_i, [c_f, c_vp, c_i, c_i, c_f, c_vp]),
    'hn_pose_side_vjp': (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_f, c_vp]),
    'hn_jacobian_vjp': (c_i, [c_f, c_f, c_i, c_i, c_i, c_f, c_vp]),
    'hn_alpha': (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_fl, c_f, c_f, c_vp]),
    'hn_composite1': (c_i, [c_f, c_f, c_f, c_f, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_vp]),
    'hn_composite2': (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_vp]),
    'hn_alpha_bwd': (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_fl, c_f, c_f, c_f, c_vp]),
    'hn_composite1_bwd': (c_i, [c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_f, c_f, c_f, c_vp]),
    'hn_composite2_bwd': (c_i, [c_f, c_f, c_f, c_f, c_f, c_f, c_i, c_i, c_f, c_f, c_f, c_f, c_vp]),
    'hn_fit_loss_sums': (c_i, [c_f, c_f, c_f, c_f, c_i, c_f, c_f, c_i, c_f, c_vp]),
    'hn_fit_loss_grads': (c_i, [c_f, c_f, c_f, c_f, c_i, c_f, c_f, c_i, c_f, c_f, c_f, c_f, c_f, c_f, c_vp]),
    'hn_adam_step': (c_i, [c_i, ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_vp), ctypes.POINTER(c_i),
                           ctypes.POINTER(c_fl), c_fl, c_fl, c_fl, ctypes.POINTER(c_i), c_vp]),
    'hn_fit_total': (c_i, [c_f, c_f, c_f, c_f, c_i, ctypes.POINTER(c_fl), c_f, c_f, c_vp]),
    'hn_fit_total_bwd': (c_i, [c_f, ctypes.POINTER(c_fl), c_f, c_f, c_f, c_i, c_f, c_f, c_f, c_f, c_vp]),
    'hn_mat3_inverse': (c_i, [c_f, c_i, c_f, c_vp]),
    'hn_mat3_inverse_bwd': (c_i, [c_f, c_f, c_i, c_f, c_vp]),
    'hn_stable_pts': (c_i, [c_f, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_vp]),
    'hn_stable_pts_bwd': (c_i, [c_f, c_i, c_i, c_i, c_f, c_f, c_f, c_vp]),
    'hn_stable_value_scratch_bytes': (c_sz, [c_i, c_i]),
    'hn_stable_value': (c_i, [c_f, c_f, c_i, c_i, c_i, c_f, c_f, c_vp, c_sz, c_vp]),
    'hn_field_tape_bytes': (c_sz, [c_vp, c_i]),
    'hn_field_eval_taped': (c_i, [c_vp, c_f, c_f, c_i, c_i, c_f, c_f, c_i, c_i, c_f, c_f, c_f, c_vp, c_sz, c_vp, c_sz, c_vp]),
    'hn_field_eval_bwd_taped': (c_i, [c_vp, c_f, c_f, c_i, c_i, c_f, c_f, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_vp, c_f, c_f, c_f, c_f, c_vp, c_sz, c_vp]),
    'hn_window_loss_scratch_bytes': (c_sz, [c_i, c_i]),
    'hn_window_loss': (c_i, [c_f, c_f, c_f, c_f, c_i, c_f, c_f, c_i, c_f, c_f, c_i, c_f, c_f, c_f, c_f, c_f, c_i, c_f, c_i, ctypes.POINTER(c_fl), c_vp, c_sz,
                             c_f, c_f, c_f, c_f, c_f, c_vp]),
    'hn_window_loss_bwd': (c_i, [c_f, c_f, c_f, c_f, c_i, c_f, c_f, c_i, c_f, c_f, ctypes.POINTER(c_fl), c_f, c_f, c_f, c_i, c_f, c_f, c_f, c_f, c_f, c_f,
                                 c_f, c_f, c_vp]),
    'hn_fit_step_loss_scratch_bytes': (c_sz, [c_i, c_i]),
    'hn_fit_step_loss': (c_i, [c_f, c_f, c_f, c_f, c_i, c_f, c_f, c_i, c_f, c_f, c_i, c_f, c_f, c_f, c_f, c_f, c_i, ctypes.POINTER(c_fl), c_vp, c_sz,
                               c_f, c_f, c_f, c_f, c_f, c_vp]),
    'hn_fit_step_loss_bwd': (c_i, [c_f, c_f, c_f, c_f, c_i, c_f, c_f, c_i, c_f, c_f, ctypes.POINTER(c_fl), c_f, c_f, c_f, c_i, c_f, c_f, c_f, c_f, c_f,
                                   c_f, c_f, c_vp]),
    'hn_variance_to_inv_s': (c_i, [c_f, c_f, c_vp]),
    'hn_variance_chain': (c_i, [c_f, c_f, c_f, c_vp]),
    'hn_train_loss': (c_i, [c_f, c_f, c_f, c_f, c_f, c_i, c_fl, c_fl, c_f, c_vp]),
    'hn_train_loss_bwd': (c_i, [c_f, c_f, c_f, c_f, c_i, c_f, c_f, c_fl, c_fl, c_f, c_f, c_f, c_vp]),
    'hn_fit_step_loss_frames': (c_i, [c_i, c_f, c_f, c_f, c_f, c_i, c_f, c_f, c_i, c_f, c_f, c_i, c_f, c_f, c_f, c_f, ctypes.POINTER(c_vp),
                                      ctypes.POINTER(c_i), ctypes.POINTER(c_fl), c_vp, c_sz, c_f, c_f, c_f, c_f, c_f, c_vp]),
    'hn_fit_step_loss_bwd_frames': (c_i, [c_i, c_f, c_f, c_f, c_f, c_i, c_f, c_f, c_i, c_f, c_f, ctypes.POINTER(c_fl), c_f, c_f, c_f, c_i, c_f, c_f, c_f,
                                          c_f, c_f, c_f, c_f, c_vp]),
    'hn_render_single_workspace_bytes': (c_sz, [c_vp, c_i, c_i, c_i]),
    'hn_render_single': (c_i, [c_vp, c_f, c_f, c_f, c_i, c_db, c_db, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_f,
                               c_f, c_f, c_vp, c_sz, c_vp]),
    'hn_render_dual_workspace_bytes': (c_sz, [c_vp, c_vp, c_i, c_i, c_i]),
    'hn_render_dual_aux_offsets': (c_i, [c_vp, c_vp, c_i, c_i, c_i, c_i, ctypes.POINTER(c_sz)]),
    'hn_render_dual_compact_offsets': (c_i, [c_vp, c_vp, c_i, c_i, c_i, c_i, ctypes.POINTER(c_sz)]),
    'hn_render_dual_bwd_workspace_bytes': (c_sz, [c_vp, c_vp, c_i, c_i]),
    'hn_render_dual_bwd': (c_i, [c_vp, c_vp, c_f, c_f, c_i, c_i, c_i, c_fl, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f,
                                 c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_vp, c_sz, c_vp, c_i, c_vp]),
    'hn_render_dual_tape_bytes': (c_sz, [c_vp, c_vp, c_i, c_i]),
    'hn_render_dual_tape_aux_offset': (c_sz, [c_vp, c_vp, c_i, c_i]),
    'hn_release_cached_memory': (c_sz, []),
    'hn_field_param_floats': (c_sz, [c_vp]),
    'hn_field_param_offset': (c_i, [c_vp, c_i, c_i, ctypes.POINTER(c_sz), ctypes.POINTER(c_sz), ctypes.POINTER(c_i),
                                    ctypes.POINTER(c_i), ctypes.POINTER(c_i)]),
    'hn_field_param_bwd': (c_i, [c_vp, c_f, c_f, c_i, c_i, c_f, c_f, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_vp, c_sz,
                                 c_vp]),
    'hn_weight_norm_bwd': (c_i, [c_vp, ctypes.POINTER(MlpDesc), ctypes.POINTER(MlpDesc), c_f, ctypes.POINTER(MlpDesc),
                                 ctypes.POINTER(MlpDesc), c_vp]),
    'hn_render_single_bwd_workspace_bytes': (c_sz, [c_vp, c_i, c_i]),
    'hn_render_single_bwd': (c_i, [c_vp, c_f, c_f, c_i, c_i, c_fl, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_vp,
                                   c_sz, c_vp]),
    'hn_render_single_tape_bytes': (c_sz, [c_vp, c_i, c_i]),
    'hn_render_single_taped': (c_i, [c_vp, c_f, c_f, c_f, c_i, c_db, c_db, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_f,
                                     c_f, c_f, c_vp, c_sz, c_vp, c_sz, c_vp]),
    'hn_render_single_bwd_taped': (c_i, [c_vp, c_f, c_f, c_i, c_i, c_fl, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_vp,
                                         c_sz, c_vp, c_sz, c_vp]),
    'hn_render_dual': (c_i, [c_vp, c_vp, c_f, c_f, c_f, c_i, c_i, c_db, c_db, c_i, c_i, c_i, c_f, c_f, c_f, c_f,
                             c_i, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_vp, c_sz, c_vp, c_sz, c_i, c_vp]),
    'hn_side_stream': (c_i, [ctypes.POINTER(c_vp)]),
    'hn_stream_wait': (c_i, [c_vp, c_vp]),
}

_lib = None


def load():
    """Load libhonerf.so (once) and set every prototype.  Raises RuntimeError if
    the library has not been built -- there is no substitute path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('libhonerf.so not found at %s: build it with `make -C %s` '
                           '(hipcc --offload-arch=gfx950); there is no CPU fallback'
                           % (LIB_PATH, os.path.join(_HERE, 'csrc')))
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    got = lib.hn_version()
    if got != HN_VERSION:           # same symbol names, different argument lists: binding them would pass garbage pointers
        raise RuntimeError('libhonerf.so at %s is ABI version %d, these bindings are for %d: rebuild it (`make -C %s`)'
                           % (LIB_PATH, got, HN_VERSION, os.path.join(_HERE, 'csrc')))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().hn_last_error()
        raise RuntimeError('%s failed (%d): %s' % (what, rc, msg.decode() if msg else ''))


def dropped_samples(reset=False):
    """hn_dropped_samples: samples the hand field's adjoint kernels dropped on the current device (out of the fp16 fragments' range, next
    to a bone's origin) since the library was loaded / the last reset.  Waits for the device."""
    n = ctypes.c_ulonglong(0)
    check(load().hn_dropped_samples(ctypes.byref(n), 1 if reset else 0), 'hn_dropped_samples')
    return int(n.value)


def ptr(t):
    """Device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), 'honerf expects contiguous device tensors'
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    """The raw handle of torch's current stream on the current device.  (Through the C bindings directly: `torch.cuda.current_stream()`
    builds a Stream object behind three Python calls, ~10 us, and a fitting step asks a dozen times.)"""
    try:
        return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))
    except Exception:
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def f32(t, device=None):
    """Contiguous fp32 device tensor view/copy of t."""
    t = t if isinstance(t, torch.Tensor) else torch.as_tensor(t)
    return t.detach().to(device=device or 'cuda', dtype=torch.float32).contiguous()
