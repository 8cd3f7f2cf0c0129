"""Synthetic weights, cameras, hand poses and ray grids.

There is no dataset and no checkpoint on the GPU box, so the bench, the smoke
test and the parity tests all draw their inputs from here.  Everything is a
pure function of a seed through ``numpy.random.RandomState`` (whose stream is
frozen across numpy versions), so the script that generates golden vectors
from the reference in the build container and the tests on the GPU box see the
same bytes without the weights having to be committed.

Shapes follow the reference confs:
  * obj nets   confs/wmask_realobj_bean.conf:39-69
  * hand nets  confs/wmask_realhand_hand1.conf:39-70
  * state-dict key layout (old-style weight-norm: ``lin{l}.weight_g``,
    ``lin{l}.weight_v``, ``lin{l}.bias``)  utils/fields.py:120-123, 216-218,
    307-309, 382-384
"""
import math

import numpy as np

N_BONES = 21
# utils/fields.py:24 -- per-bone cutoff distances of the soft bone mask h
HAND_CUTOFF = (0.08, 0.03, 0.03, 0.02, 0.02, 0.03, 0.02, 0.02, 0.02, 0.03, 0.02,
               0.02, 0.02, 0.03, 0.02, 0.02, 0.02, 0.03, 0.02, 0.02, 0.02)

D_HIDDEN = 256
D_FEATURE = 256
SDF_LAYERS = 8          # hidden layers -> 9 linear layers lin0..lin8
COLOR_LAYERS = 4        # hidden layers -> 5 linear layers lin0..lin4
SKIP_LAYER = 4
PTS_FREQS = 10          # v_multires
DIR_FREQS_OBJ = 4       # r_multires (obj conf)
DIR_FREQS_HAND = 7      # r_multires (hand conf)
GRAD_FREQS = 4          # grad_multires

OBJ_IN = 3 + 2 * 3 * PTS_FREQS                                  # 63
HAND_BONE_FEAT = (1 + 2 * PTS_FREQS) + (3 + 2 * 3 * DIR_FREQS_HAND)   # 21 + 45 = 66
HAND_IN = N_BONES * HAND_BONE_FEAT                               # 1386
GRAD_IN = 3 + 2 * 3 * GRAD_FREQS                                 # 27
OBJ_COLOR_IN = OBJ_IN + (3 + 2 * 3 * DIR_FREQS_OBJ) + D_FEATURE + GRAD_IN   # 373
HAND_COLOR_IN = HAND_IN + D_FEATURE + GRAD_IN                    # 1669


def layer_shapes(kind, d_hidden=D_HIDDEN):
    """[(out, in)] per linear layer, as the reference constructors build them.

    kind: 'sdf_obj' (utils/fields.py:271-286), 'sdf_hand' (:76-99),
          'color_obj' (:373-381), 'color_hand' (:199-214).
    """
    H = d_hidden
    if kind == 'sdf_obj':
        dims = [OBJ_IN] + [H] * SDF_LAYERS + [1 + D_FEATURE]
        shapes = []
        for l in range(len(dims) - 1):
            out = dims[l + 1] - dims[0] if (l + 1) == SKIP_LAYER else dims[l + 1]
            shapes.append((out, dims[l]))
        return shapes
    if kind == 'sdf_hand':
        dims = [HAND_IN] + [H] * SDF_LAYERS + [1 + D_FEATURE]
        shapes = []
        for l in range(len(dims) - 1):
            cin = dims[l] + dims[0] if l == SKIP_LAYER else dims[l]
            shapes.append((dims[l + 1], cin))
        return shapes
    if kind == 'color_obj':
        dims = [OBJ_COLOR_IN] + [H] * COLOR_LAYERS + [3]
        return [(dims[l + 1], dims[l]) for l in range(len(dims) - 1)]
    if kind == 'color_hand':
        dims = [HAND_COLOR_IN] + [H] * COLOR_LAYERS + [3]
        return [(dims[l + 1], dims[l]) for l in range(len(dims) - 1)]
    raise ValueError(kind)


def _freq_scale(kind):
    """Per-input-column amplitude 2^-k for a sin/cos(2^k x) column (1 for the raw
    coordinates), so every frequency band contributes O(1) to d sdf / d x."""
    def enc(channels, L):
        one = [2.0 ** -k for k in range(L)]
        return (one + one) * channels
    if kind == 'sdf_obj':
        return np.asarray([1.0] * 3 + enc(3, PTS_FREQS))
    per_bone = [1.0] + enc(1, PTS_FREQS) + [1.0] * 3 + enc(3, DIR_FREQS_HAND)
    return np.asarray(per_bone * N_BONES)


def synth_state_dict(kind, seed, d_hidden=D_HIDDEN, noise=0.1, hand_far_sdf=0.03):
    """A state dict (numpy float32) in the reference's key layout.

    The SDF nets follow the statistics of the reference's geometric init
    (utils/fields.py:100-118, 287-305: sphere-like SDF, last layer mean
    sqrt(pi)/sqrt(d), bias -0.5) but the encoding columns the reference zeroes
    are filled with noise of amplitude `noise`/2^k per frequency band and
    ``weight_g`` is detuned from ``||weight_v||``, so every input column, the
    skip path and the weight-norm fold all matter while |grad sdf| stays O(1)
    as in a trained (eikonal-regularised) field.  The hand SDF uses the
    inside-outside sign (positive far from every bone, negative near them).
    Colour nets get a plain fan-in normal init.
    """
    rng = np.random.RandomState(seed)
    shapes = layer_shapes(kind, d_hidden)
    sd = {}
    last = len(shapes) - 1
    hand = kind == 'sdf_hand'
    for l, (out, cin) in enumerate(shapes):
        if kind.startswith('sdf'):
            if l == last:
                mean = math.sqrt(math.pi) / math.sqrt(cin)
                w = rng.standard_normal((out, cin)) * 1e-4 + (-mean if hand else mean)
                # feature rows (1..256) would otherwise all equal the sdf row
                w[1:] = rng.standard_normal((out - 1, cin)) * (1.0 / math.sqrt(cin))
                b = np.full((out,), 0.08 if hand else -0.56)
                b[1:] = rng.standard_normal(out - 1) * 0.1
            else:
                std = math.sqrt(2.0) / math.sqrt(out)
                w = rng.standard_normal((out, cin)) * std
                fs = _freq_scale(kind)
                if l == 0:
                    if hand:
                        w *= 0.14 * fs[None, :]
                    else:
                        w[:, 3:] *= noise * fs[None, 3:]
                elif l == SKIP_LAYER:
                    if hand:
                        w[:, -HAND_IN:] *= 0.05 * fs[None, :]
                    else:
                        w[:, -(OBJ_IN - 3):] *= noise * fs[None, 3:]
                b = rng.standard_normal(out) * 0.01
        else:
            w = rng.standard_normal((out, cin)) * (1.0 / math.sqrt(cin))
            if l == last:
                w *= 2.0
            b = rng.standard_normal(out) * 0.1
        v = w.astype(np.float32)
        norm = np.sqrt((v.astype(np.float64) ** 2).sum(axis=1, keepdims=True))
        g = (norm * (1.0 + 0.1 * rng.standard_normal((out, 1)))).astype(np.float32)
        # effective W = g * v / ||v||, see fold_weight_norm
        sd['lin%d.weight_g' % l] = g
        sd['lin%d.weight_v' % l] = v
        sd['lin%d.bias' % l] = b.astype(np.float32)
    if hand:
        # pin the far-field value (all 1386 features exactly 0, SURVEY B-11) to a small
        # positive "outside" distance so only rays that pass a bone see a surface
        x = np.zeros(HAND_IN)
        feat = x
        for l in range(len(shapes)):
            W = fold_weight_norm(sd['lin%d.weight_g' % l], sd['lin%d.weight_v' % l]).astype(np.float64)
            if l == SKIP_LAYER:
                x = np.concatenate([x, feat]) / math.sqrt(2.0)
            x = W @ x + sd['lin%d.bias' % l]
            if l < last:
                x = np.where(x * 100.0 > 20.0, x, np.log1p(np.exp(np.minimum(x * 100.0, 20.0))) / 100.0)
        sd['lin%d.bias' % last][0] += np.float32(hand_far_sdf - x[0])
    return sd


def fold_weight_norm(g, v):
    """W[i,:] = g[i] * v[i,:] / ||v[i,:]||_2 (torch.nn.utils.weight_norm, dim=0)."""
    v = np.asarray(v, dtype=np.float32)
    n = np.sqrt((v * v).sum(axis=1, keepdims=True, dtype=np.float32))
    return (np.asarray(g, dtype=np.float32).reshape(-1, 1) * v / n).astype(np.float32)


# ----------------------------------------------------------------------------
# cameras and rays
# ----------------------------------------------------------------------------

def ring_cameras(n_views, radius=1.0, target=(0.0, 0.0, 0.0), focal=2.0, seed=0):
    """n cameras on a ring around `target`, looking at it.

    Returns dict of float32 arrays R[n,3,3], T[n,3], focal[n,2], principal[n,2]
    in the PyTorch3D row-vector convention X_view = X_world @ R + T that
    utils/utils.py:96 (cameras.unproject_points) assumes.
    """
    rng = np.random.RandomState(seed)
    target = np.asarray(target, dtype=np.float64)
    Rs, Ts = [], []
    for i in range(n_views):
        ang = 2.0 * math.pi * i / n_views + 0.1 * rng.standard_normal()
        elev = 0.3 * rng.standard_normal()
        c = target + radius * np.array([math.cos(elev) * math.sin(ang), math.sin(elev),
                                        -math.cos(elev) * math.cos(ang)])
        zax = target - c
        zax /= np.linalg.norm(zax)
        up = np.array([0.0, 1.0, 0.0])
        xax = np.cross(up, zax)
        xax /= np.linalg.norm(xax)
        yax = np.cross(zax, xax)
        R = np.stack([xax, yax, zax], axis=1)          # columns = camera axes in world
        T = -c @ R
        Rs.append(R)
        Ts.append(T)
    return {
        'R': np.asarray(Rs, dtype=np.float32),
        'T': np.asarray(Ts, dtype=np.float32),
        'focal': np.full((n_views, 2), focal, dtype=np.float32),
        'principal': np.zeros((n_views, 2), dtype=np.float32),
    }


def front_camera(dist=1.0, focal=2.0):
    """R = I, T = (0,0,dist): a camera `dist` in front of the origin."""
    return {
        'R': np.eye(3, dtype=np.float32)[None],
        'T': np.array([[0.0, 0.0, dist]], dtype=np.float32),
        'focal': np.full((1, 2), focal, dtype=np.float32),
        'principal': np.zeros((1, 2), dtype=np.float32),
    }


def ndc_grid(H, W):
    """The full-image NDC pixel grid of exp_runner.py:338-350, flattened row-major.

    x runs from +W/H to -W/H (or +1..-1), y from +1 to -1 (or +H/W..-H/W).
    Returns float32 [H*W, 2].
    """
    if W >= H:
        rx, ry = W / H, 1.0
    else:
        rx, ry = 1.0, H / W
    xs = np.linspace(rx, -rx, W, dtype=np.float32)
    ys = np.linspace(ry, -ry, H, dtype=np.float32)
    gx = np.broadcast_to(xs[None, :], (H, W)).reshape(-1)
    gy = np.broadcast_to(ys[:, None], (H, W)).reshape(-1)
    return np.stack([gx, gy], axis=-1).astype(np.float32)


def mask_pixels_ndc(H, W, n, seed):
    """n random pixels inside a synthetic elliptical mask, in the NDC convention
    of utils/dataset.py:45-47: x = -(col - W/2)/(H/2), y = -(row - H/2)/(H/2)."""
    rng = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        col = rng.randint(0, W)
        row = rng.randint(0, H)
        if ((col - W / 2) / (0.35 * W)) ** 2 + ((row - H / 2) / (0.35 * H)) ** 2 <= 1.0:
            out.append((col, row))
    cr = np.asarray(out, dtype=np.float32)
    x = -(cr[:, 0] - W / 2.0) / (H / 2.0)
    y = -(cr[:, 1] - H / 2.0) / (H / 2.0)
    return np.stack([x, y], axis=-1).astype(np.float32)


# ----------------------------------------------------------------------------
# hand pose: per-bone world -> bone-local rigid transforms
# ----------------------------------------------------------------------------

def _rodrigues(axis, ang):
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + math.sin(ang) * K + (1 - math.cos(ang)) * (K @ K)


# MANO-21 kinematic tree: parent of every joint (wrist = 0; 4 joints per finger)
HAND_PARENTS = (-1, 0, 1, 2, 3, 0, 5, 6, 7, 0, 9, 10, 11, 0, 13, 14, 15, 0, 17, 18, 19)


def synth_hand_pose(seed, center=(0.0, 0.0, 0.9), flex=0.35):
    """A plausible 21-joint hand as the renderer sees it.

    Returns (bt_inv [21,4,4], T_pose_21 [21,3], joints_world [21,3]) float32.
    bt_inv[b] maps world points into bone b's posed-to-rest frame (rigid), the
    role `bone_transformation_inv` plays in exp_runner.py:319-331; T_pose_21 is
    the rest-pose joint position, so q_b = R_b p + t_b - T_b (utils/fields.py:
    30-31) vanishes at joint b.  This is the build's own toy forward-kinematics
    chain (finger bones 2-9 cm, random flexion), not the HALO converter.
    """
    rng = np.random.RandomState(seed)
    rest = np.zeros((N_BONES, 3))
    finger_dirs = []
    for f in range(5):
        spread = (f - 2) * 0.22
        finger_dirs.append(np.array([math.sin(spread), math.cos(spread), 0.0]))
    lengths = {0: (0.09, 0.04, 0.03, 0.025), 1: (0.095, 0.045, 0.03, 0.025), 2: (0.09, 0.045, 0.03, 0.025),
               3: (0.085, 0.04, 0.028, 0.022), 4: (0.06, 0.035, 0.03, 0.025)}
    for f in range(5):
        for k in range(4):
            j = 1 + 4 * f + k
            p = HAND_PARENTS[j]
            rest[j] = rest[p] + finger_dirs[f] * lengths[f][k]
    # posed global transforms G_j (rest frame -> world)
    G = [None] * N_BONES
    root_R = _rodrigues(rng.standard_normal(3), 0.4 * rng.standard_normal())
    G[0] = (root_R, np.asarray(center, dtype=np.float64) - root_R @ rest[0])
    for j in range(1, N_BONES):
        p = HAND_PARENTS[j]
        Rp, tp = G[p]
        ang = flex * abs(rng.standard_normal()) if p != 0 else 0.15 * rng.standard_normal()
        Rl = _rodrigues(np.array([1.0, 0.0, 0.0]) + 0.1 * rng.standard_normal(3), ang)
        # rotate about the parent joint's rest position
        R = Rp @ Rl
        t = Rp @ (rest[p] - Rl @ rest[p]) + tp
        G[j] = (R, t)
    bt_inv = np.zeros((N_BONES, 4, 4))
    joints = np.zeros((N_BONES, 3))
    for j in range(N_BONES):
        R, t = G[j]
        joints[j] = R @ rest[j] + t
        bt_inv[j, :3, :3] = R.T
        bt_inv[j, :3, 3] = -R.T @ t
        bt_inv[j, 3, 3] = 1.0
    return bt_inv.astype(np.float32), rest.astype(np.float32), joints.astype(np.float32)


def synth_obj_pose(seed, center=(0.03, -0.02, 0.95)):
    """Object rotation R_obj [3,3] and translation [3] (object-local -> world)."""
    rng = np.random.RandomState(seed)
    R = _rodrigues(rng.standard_normal(3), 0.7 * rng.standard_normal())
    t = np.asarray(center, dtype=np.float64) + 0.01 * rng.standard_normal(3)
    return R.astype(np.float32), t.astype(np.float32)
