"""Shared test helpers: synthetic nets for the oracle, comparison utilities."""
import numpy as np
import torch

from honerf_amd import synth

SEEDS = {'sdf_obj': 11, 'color_obj': 12, 'sdf_hand': 21, 'color_hand': 22}
VAR_OBJ, VAR_HAND = 0.3, 0.27


def state_dicts():
    return {k: synth.synth_state_dict(k, s) for k, s in SEEDS.items()}


def oracle_fields(requires_grad=False):
    from oracle.nets import Field
    sd = state_dicts()
    obj = Field('obj', sd['sdf_obj'], sd['color_obj'], VAR_OBJ, requires_grad=requires_grad)
    hand = Field('hand', sd['sdf_hand'], sd['color_hand'], VAR_HAND, requires_grad=requires_grad)
    return hand, obj


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def rel_err(a, b):
    """max |a-b| / max(|b|) -- the 'relative fp32' measure of the north star."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def assert_close(a, b, rtol, what=''):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert a.shape == b.shape, '%s: shape %s vs %s' % (what, a.shape, b.shape)
    e = rel_err(a, b)
    assert e <= rtol, '%s: rel err %.3e > %.1e' % (what, e, rtol)


# ---- product-side helpers (GPU tests) ----------------------------------------------------
def product_modules(dev='cuda'):
    """The build's parameter containers holding the same synthetic weights as oracle_fields()."""
    from honerf_amd import nets
    m = {
        'sdf_obj': nets.SDFNetwork_OBJ(), 'color_obj': nets.RenderingNetwork_OBJ(),
        'sdf_hand': nets.SDFNetwork(), 'color_hand': nets.RenderingNetwork(use_gradients=True),
        'var_obj': nets.SingleVarianceNetwork(VAR_OBJ), 'var_hand': nets.SingleVarianceNetwork(VAR_HAND),
    }
    for k, s in SEEDS.items():
        m[k].reset_parameters(s)
    return {k: v.to(dev) for k, v in m.items()}


def packed_fields(dev='cuda'):
    from honerf_amd.nets import PackedField
    m = product_modules(dev)
    hand = PackedField('hand', m['sdf_hand'], m['color_hand'], m['var_hand'])
    obj = PackedField('obj', m['sdf_obj'], m['color_obj'], m['var_obj'])
    return hand, obj


def cu(a, dtype=None):
    x = t(a) if not isinstance(a, torch.Tensor) else a
    if dtype is not None:
        x = x.to(dtype)
    return x.cuda().contiguous()
