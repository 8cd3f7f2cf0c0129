"""ORACLE (test infrastructure, not product): the adjoint of `Field.evaluate` written out by hand.

The reference obtains d loss / d (pts, dirs, bt_inv) from autograd through the two field networks, including the
second-order path through `.gradient()` (utils/fields.py:165-177, 336-347 with create_graph=True, consumed by
fitting_single.py:289-291).  This module restates that backward pass as explicit sweeps -- no autograd -- in the
order a fused kernel runs them, so that every intermediate of the future HIP adjoint has a CPU counterpart:

  1. forward tape: z_l, a_l, s_l = sigma'(z_l), t_l = sigma''(z_l)                     (sigma = softplus, beta 100)
  2. reverse sweep (this is `.gradient()`):  u_7 = W_8[0]/scale,  dz_l = s_l * u_l,  u_{l-1} = Wh_l^T dz_l,
     GX = W_0^T dz_0 + W4x^T dz_4,  g = JX^T GX
  3. colour network forward and its ordinary backward -> adjoints of (X, enc(d), feature vector, enc(g))
  4. adjoint of step 2: GXb = JX gb (+ the input map's second-order term into pts / bone transforms), then a
     FORWARD-direction sweep  dzb_l = Wh_l (s_{l-1} * dzb_{l-1}) [+ W4x GXb at l = 4],  sb_l = u_l * dzb_l
  5. first-order reverse sweep with the extra source t_l * sb_l at every layer, down to Xb
  6. Xb through the input map's Jacobian -> pts (and, for the hand, the bone transforms)

Tested against torch.autograd on the oracle networks in float64 (tests/test_field_adjoint_spec.py).
Shapes: pts [M,3]; dirs [M,3]; hand: ONE frame, bt_inv [21,4,4], T_pose [21,3].
"""
import math

import torch

from .nets import HAND_CUTOFF, SKIP, TAU, embed

BETA = 100.0


def _softplus(z):
    return torch.nn.functional.softplus(z, beta=BETA, threshold=20.0)


def _enc_jac(x, L):
    """d [x, embed(x, L)] / dx for a [M,C] input, as the per-feature scalar derivative along its own channel:
    returns (chan [F] long, d1 [M,F], d2 [M,F]) for the layout [x (C), then per channel: sin k0..L-1, cos k0..L-1]."""
    M, C = x.shape
    freq = 2.0 ** torch.arange(L, dtype=x.dtype)
    ang = x[:, :, None] * freq                                    # [M,C,L]
    d1 = torch.stack([freq * ang.cos(), -freq * ang.sin()], dim=2)            # [M,C,2,L]
    d2 = torch.stack([-freq ** 2 * ang.sin(), -freq ** 2 * ang.cos()], dim=2)
    chan = torch.cat([torch.arange(C), torch.arange(C).repeat_interleave(2 * L)])
    d1 = torch.cat([torch.ones(M, C, dtype=x.dtype), d1.reshape(M, -1)], dim=1)
    d2 = torch.cat([torch.zeros(M, C, dtype=x.dtype), d2.reshape(M, -1)], dim=1)
    return chan, d1, d2


def _enc_pull(x, L, fbar):
    """J^T fbar for f = [x, embed(x, L)]: [M,F] -> [M,C]."""
    chan, d1, _ = _enc_jac(x, L)
    out = torch.zeros_like(x)
    out.index_add_(1, chan, d1 * fbar)
    return out


# ---------------------------------------------------------------------------------------------------------
class _ObjInput:
    """X = [p, embed(p, 10)] (utils/fields.py:318-319)."""

    def __init__(self, pts):
        self.p = pts
        self.chan, self.d1, self.d2 = _enc_jac(pts, 10)
        self.X = torch.cat([pts, embed(pts, 10)], dim=-1)

    def pull(self, Xbar, out):          # J^T Xbar -> pts
        out['g_pts'].index_add_(1, self.chan, self.d1 * Xbar)

    def grad(self, GX):                 # g = J^T GX
        g = torch.zeros_like(self.p)
        g.index_add_(1, self.chan, self.d1 * GX)
        return g

    def push(self, gbar):               # J gbar
        return self.d1 * gbar[:, self.chan]

    def second(self, GX, gbar, out):    # d/dp [(J(p)^T GX) . gbar]: every feature depends on one coordinate only
        out['g_pts'].index_add_(1, self.chan, self.d2 * GX * gbar[:, self.chan])


class _HandInput:
    """Per bone b: q = R_b p + t_b - T_b, v = |q|, r = q / v, h = 1 - sigmoid(200 (v - cutoff_b)); features
    [v, enc10(v), r, enc7(r)] * h (utils/fields.py:22-36, 134-147).  A bone's 66 features are written as
    F_f = phi_f(y_f) * h(v) with y_f one of (v, r_0, r_1, r_2)."""

    def __init__(self, pts, bt_inv, T_pose):
        self.p = pts
        self.R = bt_inv[:, :3, :3]                                     # [21,3,3]
        q = torch.einsum('bij,mj->mbi', self.R, pts) + bt_inv[:, :3, 3] - T_pose      # [M,21,3]
        self.v = q.norm(dim=-1, keepdim=True)                          # [M,21,1]
        self.r = q / self.v
        cut = torch.tensor(HAND_CUTOFF, dtype=torch.float32).to(pts.dtype).reshape(1, 21, 1)   # fp32 constants, as nets.bone_coords
        sg = torch.sigmoid(TAU * (self.v - cut))
        self.h = 1.0 - sg
        self.h1 = -TAU * sg * (1.0 - sg)                               # dh/dv
        self.h2 = -TAU * TAU * sg * (1.0 - sg) * (1.0 - 2.0 * sg)      # d2h/dv2
        M = pts.shape[0]
        y = torch.cat([self.v, self.r], dim=-1).reshape(M * 21, 4)     # the four scalar arguments of a bone
        # feature layout of one bone: [v, enc10(v), r(3), enc7(r)] -> argument index and phi, phi', phi''
        cv, d1v, d2v = _enc_jac(y[:, :1], 10)
        cr, d1r, d2r = _enc_jac(y[:, 1:], 7)
        self.arg = torch.cat([cv, cr + 1])                             # [66] in 0..3
        phi = torch.cat([y[:, :1], embed(y[:, :1], 10), y[:, 1:], embed(y[:, 1:], 7)], dim=-1)
        self.phi = phi.reshape(M, 21, 66)
        self.phi1 = torch.cat([d1v, d1r], dim=-1).reshape(M, 21, 66)
        self.phi2 = torch.cat([d2v, d2r], dim=-1).reshape(M, 21, 66)
        self.X = (self.phi * self.h).reshape(M, 21 * 66)
        self.isv = (self.arg == 0).to(pts.dtype)                       # [66] features whose argument is v

    # coefficients of one bone's scalar function F(q) = sum_f G_f F_f(q):  A(v) = h sum_{v-feat} G phi,
    # B_i(r_i) = sum_{r_i-feat} G phi;  grad F = Sv r + sum_i Sr_i (e_i - r_i r) / v
    def _coeff(self, G):
        M = G.shape[0]
        G = G.reshape(M, 21, 66)
        onehot = torch.nn.functional.one_hot(self.arg, 4).to(G.dtype)  # [66,4]
        S0 = torch.einsum('mbf,fa->mba', G * self.phi, onehot)         # sum G phi   per argument
        S1 = torch.einsum('mbf,fa->mba', G * self.phi1, onehot)        # sum G phi'
        S2 = torch.einsum('mbf,fa->mba', G * self.phi2, onehot)        # sum G phi''
        return S0, S1, S2

    def _dq(self, G):
        """grad_q F per bone [M,21,3] and the pieces the second derivative re-uses."""
        S0, S1, S2 = self._coeff(G)
        tot = S0.sum(-1, keepdim=True)                                 # sum over all features of G phi
        Sv = self.h * S1[..., :1] + self.h1 * tot
        Sr = self.h * S1[..., 1:]
        dot = (Sr * self.r).sum(-1, keepdim=True)
        return Sv * self.r + (Sr - dot * self.r) / self.v

    def grad(self, GX):                 # g = sum_b R_b^T grad_q F_b
        return torch.einsum('bij,mbi->mj', self.R, self._dq(GX))

    def _spread(self, qbar, out):
        """adjoint of q = R p + t - T per bone."""
        out['g_pts'] += torch.einsum('bij,mbi->mj', self.R, qbar)
        out['g_bt_inv'][:, :3, :3] += torch.einsum('mbi,mj->bij', qbar, self.p)
        out['g_bt_inv'][:, :3, 3] += qbar.sum(0)
        out['g_T_pose'] -= qbar.sum(0)

    def pull(self, Xbar, out):
        self._spread(self._dq(Xbar), out)

    def push(self, gbar):
        """J gbar: directional derivative of every feature along dq = R_b gbar."""
        w = torch.einsum('bij,mj->mbi', self.R, gbar)                  # [M,21,3]
        rw = (self.r * w).sum(-1, keepdim=True)
        dy = torch.cat([rw, (w - self.r * rw) / self.v], dim=-1)       # d(v, r_0..2) along w   [M,21,4]
        out = self.phi1 * self.h * dy[..., self.arg] + self.phi * self.h1 * rw
        return out.reshape(out.shape[0], -1)

    def second(self, GX, gbar, out):
        """d/d(p, R, t) [ g(p) . gbar ] with GX held fixed, g = sum_b R_b^T grad_q F_b(q_b)."""
        dq = self._dq(GX)
        out['g_bt_inv'][:, :3, :3] += torch.einsum('mbi,mj->bij', dq, gbar)        # the explicit R_b^T
        w = torch.einsum('bij,mj->mbi', self.R, gbar)
        # Hessian-vector product of F_b at q_b along w:  grad_q (grad F . w)
        S0, S1, S2 = self._coeff(GX)
        r, v, h, h1, h2 = self.r, self.v, self.h, self.h1, self.h2
        tot = S0.sum(-1, keepdim=True)
        tot1r = S1[..., 1:]                                            # dB_i/dr_i
        Sv = h * S1[..., :1] + h1 * tot
        Sr = h * tot1r
        rw = (r * w).sum(-1, keepdim=True)
        wt = (w - r * rw) / v                                          # d r along w; also grad(r.w) = wt
        # dSv along q:  dSv/dv = h1 S1v + h S2v + h2 tot + h1 S1v ;  dSv/dr_i = h1 B_i'
        dSv_dv = 2.0 * h1 * S1[..., :1] + h * S2[..., :1] + h2 * tot
        gSv = dSv_dv * r + (h1 * tot1r - (h1 * tot1r * r).sum(-1, keepdim=True) * r) / v
        # D = Sv (r.w) + sum_i Sr_i wt_i ;  grad D = gSv (r.w) + Sv wt + sum_i [grad Sr_i wt_i + Sr_i grad wt_i]
        hv = gSv * rw + Sv * wt
        # grad Sr_i = h1 B_i' r + h B_i'' (e_i - r_i r)/v
        c = (h1 * tot1r * wt).sum(-1, keepdim=True)
        e = h * S2[..., 1:] * wt
        hv = hv + c * r + (e - (e * r).sum(-1, keepdim=True) * r) / v
        # sum_i Sr_i grad wt_i,  wt_i = (w_i - r_i (r.w)) / v :
        #   grad wt_i = -[(e_i - r_i r)/v (r.w) + r_i wt] / v - wt_i r / v
        sr_r = (Sr * r).sum(-1, keepdim=True)
        sr_wt = (Sr * wt).sum(-1, keepdim=True)
        hv = hv - ((Sr - sr_r * r) / v * rw + sr_r * wt) / v - sr_wt * r / v
        self._spread(hv, out)


# ---------------------------------------------------------------------------------------------------------
def field_adjoint(field, pts, dirs, g_sdf, g_grad, g_rgb, bt_inv=None, T_pose=None):
    """Adjoint of `field.evaluate(pts, dirs, bt_inv, T_pose) -> (sdf [M,1], grad [M,3], rgb [M,3])`.

    g_sdf [M,1], g_grad [M,3], g_rgb [M,3]: upstream gradients.  Returns a dict with 'g_pts' [M,3], 'g_dirs' [M,3]
    (zero for the hand: its colour net ignores the view direction, utils/fields.py:222-240), and for the hand
    'g_bt_inv' [21,4,4], 'g_T_pose' [21,3].  Also returns the forward values under 'sdf', 'grad', 'rgb'."""
    with torch.no_grad():
        dt = pts.dtype
        sdfW = [(W.to(dt), b.to(dt)) for W, b in field.sdf]
        colW = [(W.to(dt), b.to(dt)) for W, b in field.color]
        M = pts.shape[0]
        out = {'g_pts': torch.zeros(M, 3, dtype=dt), 'g_dirs': torch.zeros(M, 3, dtype=dt)}
        if field.kind == 'obj':
            inp = _ObjInput(pts)
        else:
            inp = _HandInput(pts, bt_inv.to(dt), T_pose.to(dt))
            out['g_bt_inv'] = torch.zeros(21, 4, 4, dtype=dt)
            out['g_T_pose'] = torch.zeros(21, 3, dtype=dt)
        X = inp.X
        Din = X.shape[1]
        rs2 = 1.0 / math.sqrt(2.0)
        n = len(sdfW)                                   # 9
        H4 = sdfW[SKIP][0].shape[1] - Din               # width of the hidden part of lin4's input
        Wh = [None] + [sdfW[l][0] if l != SKIP else sdfW[l][0][:, :H4] * rs2 for l in range(1, n - 1)]
        W0, W4x, W8 = sdfW[0][0], sdfW[SKIP][0][:, H4:] * rs2, sdfW[n - 1][0]
        scale = float(field.scale)
        # 1. forward tape
        a, s, t2 = [X], [], []
        for l in range(n - 1):
            x = a[l] if l != SKIP else torch.cat([a[l], X], dim=1) * rs2
            z = x @ sdfW[l][0].T + sdfW[l][1]
            sg = torch.sigmoid(BETA * z)
            a.append(_softplus(z)); s.append(sg); t2.append(BETA * sg * (1.0 - sg))
        z8 = a[n - 1] @ W8.T + sdfW[n - 1][1]
        sdf, fvec = z8[:, :1] / scale, z8[:, 1:]
        # 2. reverse sweep
        u = [None] * (n - 1)
        dz = [None] * (n - 1)
        u[n - 2] = (W8[0] / scale).expand(M, -1)
        for l in range(n - 2, -1, -1):
            dz[l] = s[l] * u[l]
            if l > 0:
                u[l - 1] = dz[l] @ Wh[l]
        GX = dz[0] @ W0 + dz[SKIP] @ W4x
        g = inp.grad(GX)
        # 3. colour network
        gin = torch.cat([g, embed(g, 4)], dim=-1)
        if field.kind == 'obj':
            din = torch.cat([dirs, embed(dirs, 4)], dim=-1)
            cin = torch.cat([X, din, fvec, gin], dim=-1)
        else:
            cin = torch.cat([X, fvec, gin], dim=-1)
        acts, x = [cin], cin
        for l, (W, b) in enumerate(colW):
            x = x @ W.T + b
            if l < len(colW) - 1:
                x = torch.relu(x)
                acts.append(x)
        rgb = torch.sigmoid(x)
        xb = g_rgb * rgb * (1.0 - rgb)
        for l in range(len(colW) - 1, -1, -1):
            xb = xb @ colW[l][0]
            if l > 0:
                xb = xb * (acts[l] > 0).to(dt)
        if field.kind == 'obj':
            Xb_c, db, fb, gb_in = xb[:, :Din], xb[:, Din:Din + 27], xb[:, Din + 27:Din + 27 + 256], xb[:, Din + 27 + 256:]
            out['g_dirs'] = _enc_pull(dirs, 4, db)
        else:
            Xb_c, fb, gb_in = xb[:, :Din], xb[:, Din:Din + 256], xb[:, Din + 256:]
        gb = g_grad + _enc_pull(g, 4, gb_in)
        # 4. adjoint of the reverse sweep
        GXb = inp.push(gb)
        inp.second(GX, gb, out)
        sb = [None] * (n - 1)
        dzb = GXb @ W0.T
        for l in range(1, n - 1):
            sb[l - 1] = u[l - 1] * dzb
            dzb = (s[l - 1] * dzb) @ Wh[l].T
            if l == SKIP:
                dzb = dzb + GXb @ W4x.T
        sb[n - 2] = u[n - 2] * dzb
        # 5. first-order reverse sweep with the second-order sources
        ab = torch.cat([g_sdf / scale, fb], dim=1) @ W8
        Xb = Xb_c.clone()
        for l in range(n - 2, -1, -1):
            zb = s[l] * ab + t2[l] * sb[l]
            if l == SKIP:
                Xb = Xb + zb @ W4x
                ab = zb @ Wh[l]
            elif l == 0:
                Xb = Xb + zb @ W0
            else:
                ab = zb @ Wh[l]
        # 6. input map
        inp.pull(Xb, out)
        out.update(sdf=sdf, grad=g, rgb=rgb)
        return out
