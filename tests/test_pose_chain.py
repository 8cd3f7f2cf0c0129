"""The hand pose chain (fitting_single.py:206-226; halo_util/converter_fit_batch.py) -- oracle and HIP kernel against
tests/golden/pose_chain.npz, which holds what the reference's own statements produce (values and the full Jacobian, fp64)."""
import os

import numpy as np
import pytest
import torch

from helpers import record

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'pose_chain.npz')


def rel(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - b).max() / np.abs(b).max())


def test_pose_chain_oracle_matches_reference():
    from oracle.pose_chain import pose_chain
    g = np.load(GOLD)
    bt, j3, jac = pose_chain(g['ori_pose'], g['bone_len'], g['params'])
    for name, got, want in (('bt_inv', bt, g['bt_inv']), ('joint_3d', j3, g['joint_3d']), ('jacobian', jac, g['jac'])):
        e = rel(got, want)
        record('pose chain oracle (fp64) vs reference fp64: ' + name, e, 1e-9)
        assert e <= 1e-9, (name, e)   # observed 2e-12
    # initial state of the optimisation: refined joints = the predicted joints moved to ... themselves (identity refinements)
    assert np.isfinite(jac).all()


# (No finite-difference check: the chain contains the reference's stop-gradients -- the canonical transforms and the local
# coordinate systems are detached -- so its autograd Jacobian, which the fixture holds, is deliberately not the derivative
# of the values.)


@pytest.mark.gpu
def test_pose_chain_kernel_matches_reference():
    from honerf_amd.pose import PoseChainFn, pose_chain, HandPoseChain
    g = np.load(GOLD)
    dev = torch.device('cuda')
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    prm = t(g['params']).requires_grad_(True)
    bt, j3 = PoseChainFn.apply(t(g['ori_pose']), t(g['bone_len']), prm)
    floor = float(g['ref32_vs_ref64'].max())   # the reference's own fp32 run against its fp64 run: 2.7e-5
    e_bt, e_j3 = rel(bt.detach().cpu().numpy(), g['bt_inv']), rel(j3.detach().cpu().numpy(), g['joint_3d'])
    record('pose chain kernel vs reference fp64: bt_inv (the reference\'s own fp32 run: %.1e)' % floor, e_bt, 2e-5)
    record('pose chain kernel vs reference fp64: joint_3d', e_j3, 2e-5)
    assert e_bt <= 2e-5 and e_j3 <= 2e-5, (e_bt, e_j3)   # observed 6.5e-6 / 7.5e-8: the fixture's fp64 inputs rounded to fp32
    # gradients: random cotangents against the reference's Jacobian
    rng = np.random.RandomState(5)
    gb, gj = rng.standard_normal(g['bt_inv'].shape), rng.standard_normal(g['joint_3d'].shape)
    (bt * t(gb)).sum().backward(retain_graph=True)
    want = np.einsum('fok,fo->fk', g['jac'][:, :336], gb.reshape(-1, 336))
    e1 = rel(prm.grad.cpu().numpy(), want)
    prm.grad = None
    (j3 * t(gj)).sum().backward()
    want2 = np.einsum('fok,fo->fk', g['jac'][:, 336:], gj.reshape(-1, 63))
    e2 = rel(prm.grad.cpu().numpy(), want2)
    record('pose chain kernel: d/d params through bt_inv vs reference autograd', e1, 1e-5)
    record('pose chain kernel: d/d params through joint_3d vs reference autograd', e2, 1e-5)
    assert e1 <= 1e-5 and e2 <= 1e-5, (e1, e2)
    # the module with the reference's parameter names starts at the identity refinement: joint_3d = the predicted joints'
    # refined pose of case 0 (all-zero parameters)
    m = HandPoseChain(g['ori_pose'][:1], g['bone_len'][:1]).to(dev)
    b0, j0 = m()
    assert rel(b0.detach().cpu().numpy(), g['bt_inv'][:1]) <= 1e-4
    b1, _ = pose_chain(t(g['ori_pose']), t(g['bone_len']), prm[:, 0:20], prm[:, 20:27], prm[:, 27:33].reshape(-1, 3, 2), prm[:, 33:36])
    assert torch.equal(b1, bt)


@pytest.mark.gpu
def test_pose_chain_values_only_and_jacobian_only_launches_are_the_joint_launch():
    """hn_pose_chain's three forms -- values + Jacobian in one launch, values alone (jac NULL: tangents compiled out), Jacobian
    alone (bt_inv / joint_3d NULL) -- give the same bits: the fitting steps ask for values and Jacobian separately."""
    from honerf_amd import lib as L
    lib = L.load()
    g = np.load(GOLD)
    dev = torch.device('cuda')
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev).contiguous()
    ori, bl, prm = t(g['ori_pose']), t(g['bone_len']), t(g['params'])
    F = ori.shape[0]
    e = lambda *sh: torch.full(sh, float('nan'), device=dev, dtype=torch.float32)
    bt_a, j3_a, jac_a = e(F, 21, 4, 4), e(F, 21, 3), e(F, 399, 36)
    bt_b, j3_b, jac_c = e(F, 21, 4, 4), e(F, 21, 3), e(F, 399, 36)
    st = L.stream_ptr()
    L.check(lib.hn_pose_chain(L.ptr(ori), L.ptr(bl), None, L.ptr(prm), F, L.ptr(bt_a), L.ptr(j3_a), L.ptr(jac_a), st), 'joint')
    L.check(lib.hn_pose_chain(L.ptr(ori), L.ptr(bl), None, L.ptr(prm), F, L.ptr(bt_b), L.ptr(j3_b), None, st), 'values')
    L.check(lib.hn_pose_chain(L.ptr(ori), L.ptr(bl), None, L.ptr(prm), F, None, None, L.ptr(jac_c), st), 'jacobian')
    torch.cuda.synchronize()
    for name, x, y in (('bt_inv', bt_a, bt_b), ('joint_3d', j3_a, j3_b), ('jacobian', jac_a, jac_c)):
        bad = [f for f in range(F) if not torch.equal(x[f], y[f])]
        assert not bad, '%s: hands %s differ between the launch forms (max %g)' % (name, bad, float((x - y).abs().max()))
    assert not torch.isnan(jac_a).any()
    # neither outputs nor Jacobian, or only one of the value outputs: refused
    assert lib.hn_pose_chain(L.ptr(ori), L.ptr(bl), None, L.ptr(prm), F, None, None, None, st) != 0
    assert lib.hn_pose_chain(L.ptr(ori), L.ptr(bl), None, L.ptr(prm), F, L.ptr(bt_b), None, None, st) != 0


@pytest.mark.gpu
def test_fit_step_with_the_reference_pose_chain():
    """fit_step (fit type 12) over HaloPoseChain: the six refine leaves of fitting_single.py:183-198 all receive gradients
    through hn_pose_chain_bwd, the initial state reproduces the predicted joints' refined pose, and a few Adam steps run."""
    import bench
    from honerf_amd import fitting as F
    dev = torch.device('cuda')
    ren, nets, chain, views, _ = bench.build_fit(dev, 40, 1, 64, 'f16x3', halo=True)
    pose0 = chain()
    # T_pose_21 was derived from the initial state: every bone's local coordinate vanishes at its joint
    bt0, j0, tp0 = (pose0[k][0].detach().double().cpu() for k in ('bt_inv', 'joint_3d', 'T_pose_21'))    # (float64 on the host: the check itself adds no rounding)
    q = torch.einsum('bij,bj->bi', bt0[:, :3, :3], j0) + bt0[:, :3, 3] - tp0
    assert float(q.abs().max()) < 1e-5, (float(q.abs().max()), float(bt0.abs().max()), float(j0.abs().max()), float(tp0.abs().max()))
    opt = F.make_optimizer(chain, video=False)
    first = None
    for i in range(3):
        terms = F.fit_step(ren, views[i % 8], chain, opt, bench.NEAR, bench.FAR, '12')
        first = first or {k: float(v) for k, v in terms.items() if torch.is_tensor(v) and v.numel() == 1}
        assert all(np.isfinite(float(v)) for v in terms.values() if torch.is_tensor(v) and v.numel() == 1)
    moved = [float((p.detach() - p0).abs().max()) for p, p0 in zip(chain.parameters(), (torch.eye(3, device=dev)[:, :2][None], 0, torch.eye(3, device=dev)[:, :2][None], 0, 0, 0))]
    assert all(m > 0 for m in moved), moved   # every leaf took a step: its gradient arrived


@pytest.mark.gpu
def test_rigid_pose_ops_match_the_torch_form():
    """hn_rigid_pose / hn_verts_loss / hn_jacobian_vjp (RigidPoseChain on the device) against the same chain written as torch
    operators (the class's CPU form, fitting_single.py:213-217, 227-233, 119-122): values and the gradients of the step's
    pose-side loss terms w.r.t. the four rigid leaves."""
    from honerf_amd import fitting as F, synth
    rng = np.random.RandomState(3)
    bt, tp, j = synth.synth_hand_pose(7)
    R, tt = synth.synth_obj_pose(8)
    verts = (rng.standard_normal((500, 3)) * 0.02).astype(np.float32)
    rep = lambda a, n=2: np.repeat(a[None], n, 0)
    chains = {}
    for dev in ('cpu', 'cuda'):
        c = F.RigidPoseChain(rep(bt), rep(tp), rep(j), rep(R), rep(tt), verts, device=dev)
        with torch.no_grad():
            r2 = np.random.RandomState(11)
            for p in c.parameters():
                p.add_(torch.tensor(r2.standard_normal(tuple(p.shape)) * 0.05, dtype=torch.float32, device=dev))
        chains[dev] = c
    gb, gr, gt = (torch.tensor(rng.standard_normal(s), dtype=torch.float32) for s in ((2, 21, 4, 4), (2, 3, 3), (2, 3)))
    res = {}
    for dev, c in chains.items():
        pose = c()
        rnd = {'color_fine': torch.zeros(4, 3, device=dev), 'weight_sum': torch.full((4, 1), 0.5, device=dev)}
        for video in (False, True):
            terms = {}
            fused = 'obj_verts' in pose
            # the pose-side part of step_loss (the render terms are exercised elsewhere): call it through a stub render
            loss = (pose['bt_inv'] * gb.to(dev)).sum() + (pose['obj_r'] * gr.to(dev)).sum() + (pose['obj_t'] * gt.to(dev)).sum()
            from honerf_amd.fitting import pose_loss
            if fused:
                from honerf_amd.pose import VertsLossFn
                vp = VertsLossFn.apply(pose['obj_r'], pose['obj_t'], pose['Ro_pred'], pose['To_pred'], pose['obj_verts'])
                jl = pose['joint_loss'].mean() if video else pose['joint_loss'][0]
                vl = vp.mean() if video else vp[0]
                sm = VertsLossFn.apply(pose['obj_r'][1:], pose['obj_t'][1:], pose['obj_r'][:-1], pose['obj_t'][:-1], pose['obj_verts']).mean()
            else:
                jl = pose_loss(pose['joint_3d'], pose['joint3d_pred'], mean=True) if video else pose_loss(pose['joint3d_pred'][0], pose['joint_3d'][0])
                vl = (pose_loss(pose['pred_obj_v_w'], pose['compare_obj_v_w'], mean=True) if video
                      else pose_loss(pose['compare_obj_v_w'][0], pose['pred_obj_v_w'][0]))
                sm = pose_loss(pose['pred_obj_v_w'][1:], pose['pred_obj_v_w'][:-1], mean=True)
            total = loss + 30.0 * jl + 20.0 * vl + 50.0 * sm
            grads = torch.autograd.grad(total, c.parameters(), retain_graph=True)
            res[(dev, video)] = ([pose[k].detach().cpu().numpy() for k in ('bt_inv', 'joint_3d', 'obj_r', 'obj_t')], float(jl), float(vl), float(sm),
                                 [g_.cpu().numpy() for g_ in grads])
    for video in (False, True):
        a, b = res[('cuda', video)], res[('cpu', video)]
        for name, x, y in zip(('bt_inv', 'joint_3d', 'obj_r', 'obj_t'), a[0], b[0]):
            e = rel(x, y.astype(np.float64))
            record('rigid pose op vs torch form: ' + name, e, 1e-5)
            assert e <= 1e-5, (name, e)
        for name, x, y in zip(('joint loss', 'vertex loss', 'vertex smoothness'), a[1:4], b[1:4]):
            assert abs(x - y) <= 1e-5 * max(abs(y), 1e-3), (name, x, y)
        for name, x, y in zip(('obj_rot', 'obj_trans', 'palm_rot', 'palm_trans'), a[4], b[4]):
            e = rel(x, y.astype(np.float64))
            record('rigid pose op vs torch autograd: d/d %s (video=%s)' % (name, video), e, 1e-4)
            assert e <= 1e-4, (name, e)


@pytest.mark.gpu
def test_pose_adam_is_torch_adam():
    """fitting.PoseAdam (hn_adam_step: all parameter blocks, one launch) against torch.optim.Adam with the reference's six
    groups and learning rates (fitting_single.py:191-199), including a block whose .grad is None in one step."""
    from honerf_amd import fitting as F
    dev = torch.device('cuda')
    gen = torch.Generator().manual_seed(1)
    shapes = [(1, 3, 2), (1, 3), (1, 3, 2), (1, 3), (1, 20), (1, 7)]
    lrs = [5e-4, 5e-4, 5e-4, 3e-4, 1e-3, 1e-3]
    init = [torch.randn(*s, generator=gen) for s in shapes]
    pa = [torch.nn.Parameter(x.clone().to(dev)) for x in init]
    pb = [torch.nn.Parameter(x.clone().to(dev)) for x in init]
    oa = F.PoseAdam([{'params': p, 'lr': l} for p, l in zip(pa, lrs)])
    ob = torch.optim.Adam([{'params': p, 'lr': l} for p, l in zip(pb, lrs)])
    for step in range(7):
        oa.zero_grad(set_to_none=True)
        ob.zero_grad(set_to_none=True)
        for k, (a, b) in enumerate(zip(pa, pb)):
            if step == 3 and k == 4:
                continue                                             # no gradient this step: both skip the block
            g = (torch.randn(*shapes[k], generator=gen) * 10.0 ** float(torch.randint(-6, 2, (1,), generator=gen))).to(dev)
            a.grad, b.grad = g.clone(), g.clone()
        oa.step()
        ob.step()
    for k, (a, b) in enumerate(zip(pa, pb)):
        e = float((a.detach() - b.detach()).abs().max() / (b.detach() - init[k].to(dev)).abs().max())
        record('PoseAdam vs torch.optim.Adam after 7 steps, block %d (difference / movement)' % k, e, 1e-5)
        assert e <= 1e-5, (k, e)


@pytest.mark.gpu
def test_pose_adam_state_dict_interchanges_with_torch_adam():
    """PoseAdam.state_dict / load_state_dict carry torch.optim.Adam's layout: a fit checkpointed after 3 steps under one optimiser
    resumes under the other and lands where an uninterrupted torch.optim.Adam run does."""
    from honerf_amd import fitting as F
    dev = torch.device('cuda')
    gen = torch.Generator().manual_seed(5)
    shapes, lrs = [(1, 3, 2), (1, 20)], [5e-4, 1e-3]
    init = [torch.randn(*s, generator=gen) for s in shapes]
    grads = [[torch.randn(*s, generator=gen).to(dev) for s in shapes] for _ in range(6)]
    mk = lambda: [torch.nn.Parameter(x.clone().to(dev)) for x in init]
    groups = lambda ps: [{'params': p, 'lr': l} for p, l in zip(ps, lrs)]

    def run(opt, ps, steps):
        for gs in steps:
            for p, g in zip(ps, gs):
                p.grad = g.clone()
            opt.step()

    p_ref = mk()
    run(torch.optim.Adam(groups(p_ref)), p_ref, grads)
    for first, second in ((F.PoseAdam, torch.optim.Adam), (torch.optim.Adam, F.PoseAdam)):
        ps = mk()
        o1 = first(groups(ps))
        run(o1, ps, grads[:3])
        sd = o1.state_dict()
        o2 = second(groups(ps))
        o2.load_state_dict(sd)
        run(o2, ps, grads[3:])
        for k, (a, b) in enumerate(zip(ps, p_ref)):
            e = float((a.detach() - b.detach()).abs().max() / (b.detach() - init[k].to(dev)).abs().max())
            record('%s -> %s resume vs torch.optim.Adam, block %d' % (first.__name__, second.__name__, k), e, 1e-5)
            assert e <= 1e-5, (first.__name__, k, e)


@pytest.mark.gpu
def test_halo_chain_single_node_matches_the_separate_ops():
    """HaloPoseChain on the device is one autograd node (pose.HaloChainFn) -- over all frames without an index (fitting_single),
    over the window's rows of the leaves with one (fitting_video).  Both against the same chain written as separate nodes
    (PoseChainFn + RigidPoseFn joined by advanced indexing / cat / slice operators): same values, same gradients of all six
    leaves, zero gradient rows outside the window."""
    import bench
    from honerf_amd.pose import RigidPoseFn
    dev = torch.device('cuda')
    n, window = 6, [1, 2, 3, 4]
    ups = [torch.tensor(np.random.RandomState(5 + i).standard_normal(s), dtype=torch.float32, device=dev)
           for i, s in enumerate(((n, 21, 4, 4), (n, 21, 3), (n, 3, 3), (n, 3)))]

    def problem():
        chain, j, verts = bench.build_fit_data(dev, 40, n, halo=True, drift=0.003)
        with torch.no_grad():
            r2 = np.random.RandomState(11)
            for p in chain.parameters():
                p.add_(torch.tensor(r2.standard_normal(tuple(p.shape)) * 0.02, dtype=torch.float32, device=dev))
        return chain

    def separate(chain, rows):
        idx = torch.tensor(rows, device=dev)
        bt_inv, joint_3d = chain._hand(idx)
        F_ = len(rows)
        params = torch.cat([chain.obj_rot[idx].reshape(F_, 6), chain.obj_trans[idx], torch.zeros(F_, 9, device=dev)], dim=1)
        out = RigidPoseFn.apply(params, None, None, chain.Ro_pred[idx].contiguous(), chain.To_pred[idx].contiguous(), False)
        return {'bt_inv': bt_inv, 'joint_3d': joint_3d, 'obj_r': out[:, 399:408].reshape(F_, 3, 3), 'obj_t': out[:, 408:411]}

    keys = ('bt_inv', 'joint_3d', 'obj_r', 'obj_t')
    names = ('obj_rot', 'obj_trans', 'palm_rot', 'palm_trans', 'joint_refine_angle', 'palm_refine_angle')
    for rows, call in ((list(range(n)), lambda c: c()), (window, lambda c: c(window))):
        res = []
        for node in (True, False):
            chain = problem()
            pose = call(chain) if node else separate(chain, rows)
            loss = sum((pose[k] * u[rows]).sum() for k, u in zip(keys, ups))
            grads = torch.autograd.grad(loss, chain.parameters())
            res.append(([pose[k].detach().cpu().numpy() for k in keys], [g.cpu().numpy() for g in grads]))
        for x, y in zip(res[0][0], res[1][0]):
            assert np.array_equal(x, y)
        outside = [i for i in range(n) if i not in rows]
        for name, x, y in zip(names, res[0][1], res[1][1]):
            e = rel(x, y.astype(np.float64))
            record('HaloChainFn (%d of %d frames) vs separate ops: d/d %s' % (len(rows), n, name), e, 1e-6)
            assert e <= 1e-6, (name, e)
            assert not outside or float(np.abs(x[outside]).max()) == 0.0, name
