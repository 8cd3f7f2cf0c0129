"""CPU restatement of the loss blocks of the fitting scripts and of `get_stable_loss_cross`.

TEST INFRASTRUCTURE (tests/, smoke(), bench.py's cpu_baseline leg): the product never imports this.  Plain torch, written
the way the reference writes it (boolean indexing, Python loops over frames, scipy's cKDTree), so that it can serve as the
loss half of an oracle composition of a whole fitting step.  Pinned: tests/test_oracle_golden.py holds every function
here to the fixtures made by executing the reference's own statements (tests/golden/loss_single.npz, loss_video.npz,
stable_loss.npz; tests/golden/make_golden.py).
"""
import numpy as np
import torch
import torch.nn.functional as F


def rot6d_to_matrix(rot_6d):
    """utils/utils.py:11-29: columns (b1, b2, b1 x b2) of the Gram-Schmidt of the two 3-vectors."""
    r = rot_6d.reshape(-1, 3, 2)
    a1, a2 = r[:, :, 0], r[:, :, 1]
    b1 = F.normalize(a1)
    b2 = F.normalize(a2 - torch.einsum('bi,bi->b', b1, a2).unsqueeze(-1) * b1)
    b3 = torch.cross(b1, b2, dim=-1)
    return torch.stack((b1, b2, b3), dim=-1)


def pose_loss_single(target_pose, pred_pose):
    """fitting_single.py:119-122: sum of the point distances over the FIRST dimension's length."""
    cur_err = torch.norm(target_pose - pred_pose, dim=-1)
    return cur_err.sum() / cur_err.shape[0]


def pose_loss_video(target_pose, pred_pose):
    """fitting_video.py:123-126: mean point distance."""
    return torch.norm(target_pose - pred_pose, dim=-1).mean()


def interaction_terms(sdf_hand, sdf_obj):
    """contact and penetration of fitting_single.py:268-281 / fitting_video.py:294-306 from the per-sample sdfs [n,1]."""
    sdf_hand, sdf_obj = sdf_hand[:, 0], sdf_obj[:, 0]
    sdf_abs_sum = torch.abs(sdf_hand) + torch.abs(sdf_obj)
    contact_id = (sdf_abs_sum < 1e-2)
    contact_sdf = sdf_abs_sum[contact_id]
    contact_num = contact_id.float().sum() + 1e-9
    contact_loss = torch.sum(contact_sdf) / contact_num
    obj_inner_id = (sdf_obj < 0)
    hand_select_sdf = sdf_hand[obj_inner_id]
    obj_select_sdf = sdf_obj[obj_inner_id]
    penet_points_id = (hand_select_sdf < 0)
    penet_sdf = torch.abs(hand_select_sdf[penet_points_id]) + torch.abs(obj_select_sdf[penet_points_id])
    penet_num = penet_points_id.float().sum() + 1e-9
    penet_loss = torch.sum(penet_sdf) / penet_num
    return contact_loss, penet_loss


def single_step_loss(render_out, true_rgb, true_mask, joint_3d, joint3d_pred, obj_verts_loss, fit_type):
    """fitting_single.py:251-283.  render_out: color_fine [B,3], weight_sum [B,1], sdf_hand / sdf_obj [B*S,1];
    joint_3d [1,21,3]; joint3d_pred [21,3]; obj_verts_loss: the scalar of :233 (pose_loss_single(compare_obj_v_w, pred_obj_v_w))."""
    color_fine, weight_sum = render_out['color_fine'], render_out['weight_sum']
    color_error = (color_fine - true_rgb) * true_mask
    color_fine_loss = F.l1_loss(color_error, torch.zeros_like(color_error), reduction='sum') / true_mask.shape[0]
    mask_loss = F.binary_cross_entropy(weight_sum.clip(1e-3, 1.0 - 1e-3), true_mask)
    render_loss = color_fine_loss + 0.5 * mask_loss
    joint_loss = pose_loss_single(joint3d_pred, joint_3d[0])
    terms = {'color': color_fine_loss, 'mask': mask_loss, 'joint': joint_loss, 'obj_verts': obj_verts_loss}
    if fit_type == '1':
        terms['loss'] = render_loss + 100 * joint_loss + 5 * obj_verts_loss
        return terms
    pose_refine_loss = 30 * joint_loss + 20 * obj_verts_loss
    contact_loss, penet_loss = interaction_terms(render_out['sdf_hand'], render_out['sdf_obj'])
    terms['contact'], terms['penetration'] = contact_loss, penet_loss
    terms['loss'] = render_loss + (30 * contact_loss + 20 * penet_loss) + pose_refine_loss
    return terms


def video_step_loss(render_out, true_rgb, true_mask, joint_3d, joint3d_pred, pred_obj_v_w, compare_obj_v_w, index, data_num,
                    later, stable=None):
    """fitting_video.py:285-334 for one window (fit type '1234' when `stable` is given).  Tensors [F,...] over the window's 4
    frames; `later` is the reference's `iter_id + sub_iter_id + view_id > 0`."""
    color_fine, weight_sum = render_out['color_fine'], render_out['weight_sum']
    color_error = (color_fine - true_rgb) * true_mask
    color_fine_loss = F.l1_loss(color_error, torch.zeros_like(color_error), reduction='sum') / true_mask.shape[0] / true_mask.shape[1]
    mask_loss = F.binary_cross_entropy(weight_sum.clip(1e-3, 1.0 - 1e-3), true_mask)
    render_loss = 0.5 * (color_fine_loss + 0.5 * mask_loss)
    obj_verts_loss = pose_loss_video(pred_obj_v_w, compare_obj_v_w)
    joint_loss = pose_loss_video(joint_3d, joint3d_pred)
    pose_refine_loss = 30 * joint_loss + 20 * obj_verts_loss
    contact_loss, penet_loss = interaction_terms(render_out['sdf_hand'], render_out['sdf_obj'])
    interaction_loss = 30 * contact_loss + 20 * penet_loss
    smooth_loss = pose_loss_video(joint_3d[1:], joint_3d[:-1]) + pose_loss_video(pred_obj_v_w[1:], pred_obj_v_w[:-1])
    if later and int(index[0]) == 0:
        smooth_loss = smooth_loss + pose_loss_video(joint_3d[:1], joint3d_pred[:1]) + pose_loss_video(pred_obj_v_w[:1], compare_obj_v_w[:1])
    elif later and int(index[3]) == data_num - 1:
        smooth_loss = smooth_loss + pose_loss_video(joint_3d[-1:], joint3d_pred[-1:]) + pose_loss_video(pred_obj_v_w[-1:], compare_obj_v_w[-1:])
    smooth_loss = smooth_loss * 50
    loss = render_loss + interaction_loss + pose_refine_loss + smooth_loss
    terms = {'color': color_fine_loss, 'mask': mask_loss, 'joint': joint_loss, 'obj_verts': obj_verts_loss, 'contact': contact_loss,
             'penetration': penet_loss, 'smooth': smooth_loss}
    if stable is not None:
        terms['stable'] = stable * 100
        loss = loss + terms['stable']
    terms['loss'] = loss
    return terms


def stable_loss_cross(hand_sdf_fn, pts, bt_inv, T_pose_21, Ro, To):
    """`get_stable_loss_cross` (utils/renderer_batch.py:318-371), statement by statement.  hand_sdf_fn(pts_world [F,V,3], bt_inv,
    T_pose_21) -> [F*V,1] stands for `self.sdf_network_hand.sdf`.  Note :349: `np.setdiff1d(vert_id_all, cur_in_id)` is applied to
    the boolean MASK (quirk B-12 of DESIGN.md), reproduced here because numpy does the same thing to the same arguments."""
    from scipy import spatial
    pts = pts[:, ::10, :]
    batch_size, p_num, _ = pts.shape
    pts_world = (Ro.unsqueeze(1) @ pts.unsqueeze(-1))[..., 0] + To.unsqueeze(1)
    vert_id_all = range(p_num)
    hand_sdf = hand_sdf_fn(pts_world, bt_inv, T_pose_21).reshape(batch_size, p_num, 1)
    hand_sdf_list, in_id_list = [], []
    for batch_id in range(batch_size):
        cur_hand_sdf = hand_sdf[batch_id].reshape(-1)
        penet_id = (cur_hand_sdf < 0)
        if penet_id.float().sum() > 0:
            in_id_list.append(penet_id)
            hand_sdf_list.append(cur_hand_sdf)
    stable_loss = 0
    if len(in_id_list) > 1:
        hand_sdf_list = torch.stack(hand_sdf_list, 0)
        in_time = hand_sdf_list.shape[0]
        for cid in range(in_time):
            cur_in_id = in_id_list[cid].clone().cpu()
            cur_out_id = np.setdiff1d(vert_id_all, cur_in_id)
            in_points = pts[0, cur_in_id].clone().detach().cpu()
            out_points = pts[0, cur_out_id].clone().detach().cpu()
            in_points_num = in_points.shape[0]
            nn_index = spatial.cKDTree(out_points)
            _, near_out_id = nn_index.query(in_points, k=1)
            near_out_id = np.unique(near_out_id.reshape(-1))
            in_err = hand_sdf_list[:, cur_in_id].clip(0, 1e7).sum() / ((in_time - 1) * in_points_num)
            hand_sdf_select = hand_sdf_list[:, cur_out_id]
            out_err = torch.abs(hand_sdf_select[:, near_out_id].clip(-1e7, 0)).sum() / ((in_time - 1) * in_points_num)
            stable_loss = stable_loss + in_err + 0.05 * out_err
        stable_loss = stable_loss / in_time
    return stable_loss
