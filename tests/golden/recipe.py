"""Frozen-stream recipes shared by gen_golden.py (build container, reference imported)
and the tests (anywhere).  np.random.RandomState is a frozen legacy stream, so the
same seed gives the same weights/inputs on every machine: full-size golden cases
store only outputs + gradient digests, and weights are regenerated from the seed.
"""
import numpy as np


def recipe_value(rs, name, shape):
    """One tensor of the recipe; consumed in sorted-name order."""
    shape = tuple(shape)
    v = rs.standard_normal(shape if shape else (1,)).reshape(shape)
    leaf = name.split(".")[-1]
    if leaf == "running_var":
        v = np.abs(v) * 0.5 + 0.5
    elif leaf == "running_mean":
        v = v * 0.1
    elif leaf == "weight_g":
        v = np.abs(v) * 0.5 + 0.5
    elif len(shape) >= 2:
        v = v / np.sqrt(float(np.prod(shape[1:])))
    elif leaf == "weight":          # 1-D weight = a norm layer's gamma
        v = 1.0 + 0.1 * v
    else:
        v = 0.1 * v
    return v.astype(np.float32)


def fill_by_shapes(shapes, seed):
    """name->shape table -> name->float32 array, identical to fill_module on a module
    whose floating parameters/buffers have exactly these names and shapes."""
    rs = np.random.RandomState(seed)
    return {n: recipe_value(rs, n, shapes[n]) for n in sorted(shapes)}


def _unique_named_tensors(module):
    items = list(module.named_parameters()) + list(module.named_buffers())
    return sorted(((n, t) for n, t in items if t.dtype.is_floating_point), key=lambda kv: kv[0])


def fill_module(module, seed):
    """Overwrite every floating parameter/buffer of `module` in sorted-name order."""
    import torch
    rs = np.random.RandomState(seed)
    with torch.no_grad():
        for name, t in _unique_named_tensors(module):
            t.copy_(torch.from_numpy(recipe_value(rs, name, tuple(t.shape))).to(t.device))
    return module


def named_tensors(module):
    return {n: t.detach().cpu().numpy().copy() for n, t in _unique_named_tensors(module)}


def draw(rs, shape, kind="normal"):
    if kind == "normal":
        return rs.standard_normal(shape).astype(np.float32)
    if kind == "uniform_pm1":
        return rs.uniform(-1, 1, shape).astype(np.float32)
    raise ValueError(kind)


def grad_digest(g):
    """Compact fingerprint of a gradient array: L2 norm, sum, first 8 values."""
    g = np.asarray(g, np.float64).reshape(-1)
    head = np.zeros(8)
    head[:min(8, g.size)] = g[:8]
    return np.concatenate([[np.sqrt((g * g).sum()), g.sum()], head])


# ---- name->shape tables of the reference modules (checkpoint contract) ----
def gru_shapes(prefix, I, H, L, nC, nFC=1, dropout=False):
    s = {}
    for l in range(L):
        for sfx in ("", "_reverse"):
            s["%sgru.weight_ih_l%d%s" % (prefix, l, sfx)] = (3 * H, I if l == 0 else 2 * H)
            s["%sgru.weight_hh_l%d%s" % (prefix, l, sfx)] = (3 * H, H)
            s["%sgru.bias_ih_l%d%s" % (prefix, l, sfx)] = (3 * H,)
            s["%sgru.bias_hh_l%d%s" % (prefix, l, sfx)] = (3 * H,)
    if nC > 0:
        if nFC == 1:
            s[prefix + "fc.weight"], s[prefix + "fc.bias"] = (nC, 2 * H), (nC,)
        else:
            step = 3 if dropout else 2
            dims = [2 * H] + [H] * (nFC - 1) + [nC]
            for i in range(nFC):
                s["%sfc.%d.weight" % (prefix, i * step)] = (dims[i + 1], dims[i])
                s["%sfc.%d.bias" % (prefix, i * step)] = (dims[i + 1],)
    return s


def tcn_shapes(prefix, num_inputs, channels, k):
    s = {}
    for i, co in enumerate(channels):
        ci = num_inputs if i == 0 else channels[i - 1]
        for c, cin in (("conv1", ci), ("conv2", co)):
            s["%snetwork.%d.%s.weight_v" % (prefix, i, c)] = (co, cin, k)
            s["%snetwork.%d.%s.weight_g" % (prefix, i, c)] = (co, 1, 1)
            s["%snetwork.%d.%s.bias" % (prefix, i, c)] = (co,)
        if ci != co:
            s["%snetwork.%d.downsample.weight" % (prefix, i)] = (co, ci, 1)
            s["%snetwork.%d.downsample.bias" % (prefix, i)] = (co,)
    return s


def simple_tcn_shapes(in_dim, k, n_out, hidden=512):
    """ModuleList([Sequential(Conv1d, BN1d, ReLU, Conv1d, BN1d, ReLU), Linear]) of the `tcn_simple`
    back-end (reference models/backbone.py:106-113, 212-238); floating tensors only."""
    s = {}
    for ci, bi, cin in ((0, 1, in_dim), (3, 4, hidden)):
        s["0.%d.weight" % ci] = (hidden, cin, k)
        s["0.%d.bias" % ci] = (hidden,)
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            s["0.%d.%s" % (bi, leaf)] = (hidden,)
    s["1.weight"] = (n_out, hidden)
    s["1.bias"] = (n_out,)
    return s


def att_fusion_shapes(prefix, dims, hidden):
    s = {}
    if dims[0] != dims[1]:
        s[prefix + "proj_v.weight"], s[prefix + "proj_v.bias"] = (dims[0], dims[1]), (dims[0],)
    s.update(gru_shapes(prefix + "scorer_a.", dims[0], hidden, 1, 1, 1))
    s.update(gru_shapes(prefix + "scorer_v.", dims[0], hidden, 1, 1, 1))
    return s


def c3_param_shapes(d_a=128, d_v=256, nh=512):
    s = {}
    s.update(gru_shapes("audio.", d_a, 256, 2, -1, 2))
    s.update(gru_shapes("visual.gru_v.", d_v, nh, 2, -1, 2))
    s.update(gru_shapes("visual.gru_a.", d_v, nh, 2, -1, 2))
    s["proj_v.weight"], s["proj_v.bias"] = (512, nh * 4), (512,)
    s.update(att_fusion_shapes("att_fuse.", [512, 512], 128))
    s.update(gru_shapes("fusion.", 512, nh, 2, 9, 2))
    return s
