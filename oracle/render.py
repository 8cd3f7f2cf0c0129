"""ORACLE (test infrastructure, not product): CPU restatement of HO-NeRF's
NeuS-style renderers in plain PyTorch fp32.  See oracle/nets.py for the rules.

All randomness is passed in (`t_rand`, one U[0,1) number per ray, the only RNG
draw on the path: utils/renderer.py:210-212); every function returns the
intermediate integer indices too so tests can demand bit-exact agreement.
"""
import torch
import torch.nn.functional as F

from .nets import as_t


# ----------------------------------------------------------------------------
# cameras / rays
# ----------------------------------------------------------------------------

def unproject_ndc(xy, depth, R, T, focal, principal):
    """PyTorch3D PerspectiveCameras.unproject_points(from_ndc=True), restated
    from its documented convention (third-party, unpinned: SURVEY 8c).
    X_view = ((x-px) z / fx, (y-py) z / fy, z);  X_world = (X_view - T) R^T.
    xy [B,2], depth scalar, R [3,3], T [3], focal [2], principal [2] -> [B,3]"""
    z = torch.full_like(xy[:, :1], depth)
    xv = torch.cat([(xy[:, 0:1] - principal[0]) * z / focal[0],
                    (xy[:, 1:2] - principal[1]) * z / focal[1], z], dim=-1)
    return (xv - T) @ R.transpose(0, 1)


def rays_from_xy(xy, R, T, focal, principal):
    """_xy_to_ray_bundle, utils/utils.py:79-108: unproject at depth 1 and 2,
    d = normalize(p2 - p1), o = p1 - d."""
    p1 = unproject_ndc(xy, 1.0, R, T, focal, principal)
    p2 = unproject_ndc(xy, 2.0, R, T, focal, principal)
    d = F.normalize(p2 - p1, dim=-1)
    return p1 - d, d


def obj_local(rays_o, rays_d, Ro, To, repeat=False):
    """convert_obj_to_local: o' = Ro (o - To), d' = Ro d.
    utils/renderer.py:180-188 (single field: Ro broadcast as [1,3,3]),
    :424-432 (fitting: Ro repeated to [B,3,3] -> `repeat=True`; torch.matmul
    picks different kernels for the two forms once Ro requires grad, so the
    distinction is kept), utils/renderer_batch.py:176-182 ([F,P,3] rays with
    Ro [F,3,3], To [F,3])."""
    if rays_o.dim() == 3:
        o = rays_o - To[:, None, :]
        return (torch.matmul(Ro[:, None], o[..., None])[..., -1],
                torch.matmul(Ro[:, None], rays_d[..., None])[..., -1])
    R = Ro[None].repeat(rays_o.shape[0], 1, 1) if repeat else Ro[None]
    o = rays_o - To[None, :]
    return (torch.matmul(R, o[..., None])[..., -1],
            torch.matmul(R, rays_d[..., None])[..., -1])


def coarse_z(near, far, n_samples, t_rand):
    """utils/renderer.py:204-212.  t_rand [...,1] in [0,1) -> z [...,n]."""
    sample_dist = (far - near) / n_samples
    z = near + (far - near) * torch.linspace(0.0, 1.0, n_samples)
    return z + (as_t(t_rand) - 0.5) * sample_dist


# ----------------------------------------------------------------------------
# hierarchical sampling
# ----------------------------------------------------------------------------

def sample_pdf_det(bins, weights, n_new):
    """sample_pdf(det=True), utils/renderer.py:10-37.  bins [B,k], weights
    [B,k-1] -> samples [B,n_new], inds int64 [B,n_new] (searchsorted result)."""
    weights = weights + 1e-5
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    u = torch.linspace(0.5 / n_new, 1.0 - 0.5 / n_new, steps=n_new)
    u = u.expand(list(cdf.shape[:-1]) + [n_new]).contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    cdf_lo, cdf_hi = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    bin_lo, bin_hi = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = cdf_hi - cdf_lo
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_lo) / denom
    return bin_lo + t * (bin_hi - bin_lo), inds


def upsample_weights(z, sdf, inv_s):
    """The NeuS section-slope alpha/weights of up_sample, utils/renderer.py:
    64-83.  z, sdf [B,k] -> weights [B,k-1]."""
    prev_sdf, next_sdf = sdf[:, :-1], sdf[:, 1:]
    prev_z, next_z = z[:, :-1], z[:, 1:]
    mid_sdf = (prev_sdf + next_sdf) * 0.5
    cos = (next_sdf - prev_sdf) / (next_z - prev_z + 1e-5)
    prev_cos = torch.cat([torch.zeros_like(cos[:, :1]), cos[:, :-1]], dim=-1)
    cos = torch.minimum(prev_cos, cos).clip(-1e3, 0.0)
    dist = next_z - prev_z
    prev_cdf = torch.sigmoid((mid_sdf - cos * dist * 0.5) * inv_s)
    next_cdf = torch.sigmoid((mid_sdf + cos * dist * 0.5) * inv_s)
    alpha = (prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)
    trans = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-7], -1), -1)[:, :-1]
    return alpha * trans


def up_sample(z, sdf, n_new, inv_s):
    """NeuSRenderer.up_sample, utils/renderer.py:60-86 -> (z_new [B,n_new], inds)."""
    return sample_pdf_det(z, upsample_weights(z, sdf, inv_s), n_new)


def merge_z(z, z_new, sdf=None, sdf_new=None):
    """cat_z_vals, utils/renderer.py:88-105: cat + sort; SDF permuted by the
    sort index.  -> (z_sorted, sdf_sorted or None, index int64)."""
    zc = torch.cat([z, z_new], dim=-1)
    zs, index = torch.sort(zc, dim=-1)
    if sdf is None:
        return zs, None, index
    sc = torch.cat([sdf, sdf_new], dim=-1)
    return zs, torch.gather(sc, 1, index), index


def merge_z_batch_quirk(z, z_new, sdf, sdf_new):
    """Batched cat_z_vals, utils/renderer_batch.py:96-113, INCLUDING its row
    indexing quirk (SURVEY B-1): the gather row is arange(P) for every frame,
    so frames 1.. pick their SDF values from frame 0's rows.  z [F,P,k]."""
    Fr, P, k = z.shape
    m = z_new.shape[-1]
    zc = torch.cat([z, z_new], dim=-1)
    zs, index = torch.sort(zc, dim=-1)
    sc = torch.cat([sdf, sdf_new], dim=-1).reshape(Fr * P, k + m)
    rows = torch.arange(P)[None, :, None].expand(Fr, P, k + m).reshape(-1)
    out = sc[(rows, index.reshape(-1))].reshape(Fr, P, k + m)
    return zs, out, index


# ----------------------------------------------------------------------------
# SDF -> alpha, compositing
# ----------------------------------------------------------------------------

def mid_points(z, sample_dist):
    """utils/renderer.py:119-121: dists (last = sample_dist), mid z."""
    d = z[..., 1:] - z[..., :-1]
    d = torch.cat([d, torch.full_like(d[..., :1], sample_dist)], -1)
    return z + d * 0.5, d


def sdf_to_alpha(sdf, grad, dirs, dists, inv_s):
    """utils/renderer.py:147-161 with cos_anneal_ratio = 1.
    sdf [N,1], grad/dirs [N,3], dists [N,1] -> alpha [N,1] (clipped), c [N,1]."""
    true_cos = (dirs * grad).sum(-1, keepdim=True)
    iter_cos = -F.relu(-true_cos)
    nxt = sdf + iter_cos * dists * 0.5
    prv = sdf - iter_cos * dists * 0.5
    c = torch.sigmoid(prv * inv_s)
    p = c - torch.sigmoid(nxt * inv_s)
    return ((p + 1e-5) / (c + 1e-5)).clip(0.0, 1.0), c


def composite_single(alpha, c, rgb):
    """utils/renderer.py:163-164: transmittance seeded with c_0 (SURVEY B-3).
    alpha, c [B,S]; rgb [B,S,3] -> weights [B,S], colour [B,3]."""
    w = alpha * torch.cumprod(torch.cat([c[:, :1], 1.0 - alpha + 1e-7], -1), -1)[:, :-1]
    return w, (rgb * w[:, :, None]).sum(dim=1)


def composite_dual(alpha_h, rgb_h, alpha_o, rgb_o):
    """utils/renderer.py:512-524 (works on [...,S] / [...,S,3])."""
    fa = (1.0 - alpha_h + 1e-7) * (1.0 - alpha_o + 1e-7)
    T = torch.cumprod(torch.cat([torch.ones_like(fa[..., :1]), fa], -1), -1)[..., :-1]
    wh, wo = alpha_h * T, alpha_o * T
    color = (rgb_h * wh[..., None]).sum(dim=-2) + (rgb_o * wo[..., None]).sum(dim=-2)
    wsum = wh.sum(dim=-1, keepdim=True) + wo.sum(dim=-1, keepdim=True)
    return color, wsum, wh, wo


def eikonal(grad, shape):
    """utils/renderer.py:166-169: mean((||g|| - 1)^2) over every sample."""
    g = grad.reshape(*shape, 3)
    return ((torch.linalg.norm(g, ord=2, dim=-1) - 1.0) ** 2).mean()


# ----------------------------------------------------------------------------
# whole renders
# ----------------------------------------------------------------------------

def _pts(o, d, z):
    return o[..., None, :] + d[..., None, :] * z[..., :, None]


def render_single(field, rays_o, rays_d, near, far, t_rand, n_samples, n_importance,
                  up_sample_steps=4, bt_inv=None, T_pose=None, Ro=None, To=None):
    """NeuSRenderer.render + render_core, utils/renderer.py:190-258, 107-177.
    `field` is an oracle.nets.Field.  Returns the reference's dict plus
    'z_vals', 'weights', 'inds' (list per step), 'index' (list per step)."""
    if field.kind == 'obj':
        rays_o, rays_d = obj_local(rays_o, rays_d, Ro, To)
    B = rays_o.shape[0]
    sample_dist = (far - near) / n_samples
    z = coarse_z(near, far, n_samples, as_t(t_rand).reshape(B, 1))
    inds_all, index_all = [], []
    if n_importance > 0:
        with torch.no_grad():
            sdf = field.sdf_only(_pts(rays_o, rays_d, z).reshape(-1, 3), bt_inv, T_pose).reshape(B, n_samples)
            for i in range(up_sample_steps):
                z_new, inds = up_sample(z, sdf, n_importance // up_sample_steps, 64 * 2 ** i)
                inds_all.append(inds)
                if i + 1 == up_sample_steps:
                    z, _, index = merge_z(z, z_new)
                else:
                    sdf_new = field.sdf_only(_pts(rays_o, rays_d, z_new).reshape(-1, 3), bt_inv, T_pose)
                    z, sdf, index = merge_z(z, z_new, sdf, sdf_new.reshape(B, -1))
                index_all.append(index)
    S = z.shape[1]
    mid_z, dists = mid_points(z, sample_dist)
    pts = _pts(rays_o, rays_d, mid_z).reshape(-1, 3)
    dirs = rays_d[:, None, :].expand(B, S, 3).reshape(-1, 3)
    sdf, grad, rgb = field.evaluate(pts, dirs, bt_inv, T_pose)
    inv_s = field.inv_s()
    alpha, c = sdf_to_alpha(sdf, grad, dirs, dists.reshape(-1, 1), inv_s)
    alpha, c = alpha.reshape(B, S), c.reshape(B, S)
    w, color = composite_single(alpha, c, rgb.reshape(B, S, 3))
    return {
        'color_fine': color,
        's_val': (1.0 / inv_s).expand(B, S).mean(dim=-1, keepdim=True),
        'cdf_fine': c,
        'weight_sum': w.sum(dim=-1, keepdim=True),
        'weight_max': torch.max(w, dim=-1, keepdim=True)[0],
        'gradient_error': eikonal(grad, (B, S)),
        'z_vals': z, 'weights': w, 'inds': inds_all, 'index': index_all,
        'sdf': sdf, 'gradients': grad, 'alpha': alpha, 'rgb': rgb.reshape(B, S, 3),
    }


def _alpha_sample_color(field, o, d, z, sample_dist, bt_inv, T_pose, batched):
    """get_alpha_sample_color, utils/renderer.py:360-422 / renderer_batch.py:115-174."""
    lead = z.shape[:-1]
    S = z.shape[-1]
    mid_z, dists = mid_points(z, sample_dist)
    pts = _pts(o, d, mid_z)
    dirs = d[..., None, :].expand(pts.shape).reshape(-1, 3)
    if field.kind == 'hand' and batched:
        sdf, grad, rgb = field.evaluate(pts.reshape(lead[0], -1, 3), dirs, bt_inv, T_pose)
    else:
        sdf, grad, rgb = field.evaluate(pts.reshape(-1, 3), dirs, bt_inv, T_pose)
    alpha, _ = sdf_to_alpha(sdf, grad, dirs, dists.reshape(-1, 1), field.inv_s())
    return (alpha.reshape(*lead, S), rgb.reshape(*lead, S, 3), sdf.reshape(-1, 1),
            eikonal(grad, (*lead, S)), grad.reshape(-1, 3))


def render_dual(hand, obj, rays_o, rays_d, near, far, t_rand, n_samples, n_importance,
                up_sample_steps, bt_inv, T_pose, Ro, To, batch_quirk=True):
    """NeuSRenderer_fitting.render: utils/renderer.py:434-535 for rays [B,3],
    utils/renderer_batch.py:184-281 for rays [F,P,3] (bt_inv [F,21,4,4],
    T_pose [F,21,3] or [21,3], Ro [F,3,3], To [F,3]).  batch_quirk reproduces
    SURVEY B-1 in the batched up-sampling."""
    batched = rays_o.dim() == 3
    o_h, d_h = rays_o, rays_d
    o_o, d_o = obj_local(rays_o, rays_d, Ro, To, repeat=True)
    lead = rays_o.shape[:-1]
    sample_dist = (far - near) / n_samples
    z = coarse_z(near, far, n_samples, as_t(t_rand).reshape(*lead, 1))
    dbg = {'inds_hand': [], 'inds_obj': []}
    if n_importance > 0:
        with torch.no_grad():
            def sdf_at(field, o, d, zz):
                p = _pts(o, d, zz)
                if field.kind == 'hand' and batched:
                    s = field.sdf_only(p.reshape(lead[0], -1, 3), bt_inv, T_pose)
                else:
                    s = field.sdf_only(p.reshape(-1, 3), bt_inv, T_pose)
                return s.reshape(*lead, zz.shape[-1])

            z_h, z_o = z, z
            s_h, s_o = sdf_at(hand, o_h, d_h, z), sdf_at(obj, o_o, d_o, z)
            n_new = n_importance // up_sample_steps
            for i in range(up_sample_steps):
                last = i + 1 == up_sample_steps
                news = []
                for field, o, d, zz, ss, key in ((hand, o_h, d_h, z_h, s_h, 'inds_hand'),
                                                 (obj, o_o, d_o, z_o, s_o, 'inds_obj')):
                    k = zz.shape[-1]
                    z_new, inds = up_sample(zz.reshape(-1, k), ss.reshape(-1, k), n_new, 64 * 2 ** i)
                    z_new = z_new.reshape(*lead, n_new)
                    dbg[key].append(inds)
                    if last:
                        zz2, ss2 = torch.sort(torch.cat([zz, z_new], -1), dim=-1)[0], ss
                    else:
                        s_new = sdf_at(field, o, d, z_new)
                        if batched and batch_quirk:
                            zz2, ss2, _ = merge_z_batch_quirk(zz, z_new, ss, s_new)
                        else:
                            zz2, ss2, _ = merge_z(zz.reshape(-1, k), z_new.reshape(-1, n_new),
                                                  ss.reshape(-1, k), s_new.reshape(-1, n_new))
                            zz2, ss2 = zz2.reshape(*lead, -1), ss2.reshape(*lead, -1)
                    news.append((zz2, ss2, z_new))
                (z_h, s_h, new_h), (z_o, s_o, new_o) = news
                z = torch.cat([z, new_h, new_o], dim=-1)
    z, _ = torch.sort(z, dim=-1)
    a_h, c_h, sdf_h, ge_h, g_h = _alpha_sample_color(hand, o_h, d_h, z, sample_dist, bt_inv, T_pose, batched)
    a_o, c_o, sdf_o, ge_o, g_o = _alpha_sample_color(obj, o_o, d_o, z, sample_dist, bt_inv, T_pose, batched)
    color, wsum, wh, wo = composite_dual(a_h, c_h, a_o, c_o)
    return {
        'color_fine': color, 'weight_sum': wsum,
        'sdf_hand': sdf_h, 'sdf_obj': sdf_o,
        'gradient_error_hand': ge_h, 'gradient_error_obj': ge_o,
        'gradient_hand': g_h, 'gradient_obj': g_o,
        'z_vals': z, 'alpha_hand': a_h, 'alpha_obj': a_o, 'rgb_hand': c_h, 'rgb_obj': c_o,
        **dbg,
    }
